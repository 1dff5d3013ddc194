"""FID / Inception-score arithmetic and dataset statistics (host, float64).

Replaces the parts of cld_jax/evaluation.py and run_lib.check_fid (cld_jax/run_lib.py:735-791) that do not need the TF-Hub
Inception network: `load_dataset_stats` (evaluation.py:45-60), and the two tensorflow_gan reductions the reference calls,
`tfgan.eval.frechet_classifier_distance_from_activations(data_pools, all_pools)` and
`tfgan.eval.classifier_score_from_logits(all_logits)` (run_lib.py:776-780).  tensorflow_gan is not vendored in the reference;
its published definitions are restated here:

  FID  = |m - m_w|^2 + tr(S + S_w) - 2 tr( (S^1/2 S_w S^1/2)^1/2 ),  S = unbiased covariance of the pool_3 activations,
         activations promoted to float64 first;
  IS   = exp( mean_x KL( p(y|x) || p(y) ) ),  p(y|x) = softmax(logits), p(y) its batch mean.

The Inception features themselves come from `tfhub.dev/tensorflow/tfgan/eval/inception/1` in the reference
(evaluation.py:25-43): an external asset that cannot be fetched here.  `run_lib.check_fid` therefore takes the feature
extractor as a callable and otherwise reuses `statistics_<r>.npz` files written by any tool that has the network.
"""
import os

import numpy as np


def load_dataset_stats(config, root=None):
    """evaluation.py:45-60: the .npz with key 'pool_3' under assets/stats/ (relative to `root`, default the working dir)."""
    ds = config.data.dataset
    if ds == 'CIFAR10':
        filename = 'assets/stats/cifar10_stats.npz'
    elif ds == 'CELEBA':
        filename = 'assets/stats/celeba_64_stats.npz'
    elif ds == 'LSUN':
        filename = 'assets/stats/lsun_%s_%s_stats.npz' % (config.data.category, config.data.image_size)
    elif ds == 'ImageNet':
        filename = 'assets/stats/imagenet%s_stats.npz' % config.data.image_size
    else:
        raise ValueError('Dataset %s stats not found.' % ds)
    return np.load(os.path.join(root or os.getcwd(), filename))


def _sym_sqrt(m):
    """Square root of a symmetric positive semi-definite matrix (negative round-off eigenvalues clamped to 0)."""
    w, v = np.linalg.eigh((m + m.T) / 2)
    return (v * np.sqrt(np.clip(w, 0.0, None))) @ v.T


def trace_sqrt_product(sigma, sigma_v):
    """tr( sqrt( sqrt(sigma) sigma_v sqrt(sigma) ) ) = sum of the square roots of the eigenvalues of sigma sigma_v."""
    s = _sym_sqrt(sigma)
    w = np.linalg.eigvalsh((s @ sigma_v @ s + (s @ sigma_v @ s).T) / 2)
    return float(np.sqrt(np.clip(w, 0.0, None)).sum())


def frechet_classifier_distance_from_activations(real_activations, generated_activations):
    real = np.asarray(real_activations, dtype=np.float64)
    gen = np.asarray(generated_activations, dtype=np.float64)
    assert real.ndim == 2 and gen.ndim == 2 and real.shape[1] == gen.shape[1]
    m, m_w = real.mean(axis=0), gen.mean(axis=0)
    sigma = np.cov(real, rowvar=False, ddof=1)
    sigma_w = np.cov(gen, rowvar=False, ddof=1)
    sigma, sigma_w = np.atleast_2d(sigma), np.atleast_2d(sigma_w)
    trace = np.trace(sigma + sigma_w) - 2.0 * trace_sqrt_product(sigma, sigma_w)
    return float(trace + np.sum((m - m_w) ** 2))


def classifier_score_from_logits(logits):
    logits = np.asarray(logits, dtype=np.float64)
    z = logits - logits.max(axis=1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
    p = np.exp(logp)
    log_marginal = np.log(p.mean(axis=0, keepdims=True))
    kl = (p * (logp - log_marginal)).sum(axis=1)
    return float(np.exp(kl.mean()))
