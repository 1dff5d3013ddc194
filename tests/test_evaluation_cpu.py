"""FID / IS arithmetic (gddim_amd/evaluation.py) against closed forms and scipy, and the report plumbing of run_lib.check_fid."""
import numpy as np
import pytest
import scipy.linalg

from gddim_amd import evaluation as ev
from gddim_amd import run_lib
from gddim_amd.config_dict import ConfigDict


def test_frechet_distance_matches_sqrtm_and_closed_forms():
    rng = np.random.default_rng(0)
    a = rng.standard_normal((400, 6)) @ rng.standard_normal((6, 6)) + 1.0
    b = rng.standard_normal((300, 6)) @ rng.standard_normal((6, 6)) - 0.5
    got = ev.frechet_classifier_distance_from_activations(a, b)
    sa, sb = np.cov(a, rowvar=False), np.cov(b, rowvar=False)
    want = np.sum((a.mean(0) - b.mean(0)) ** 2) + np.trace(sa + sb) - 2 * np.trace(scipy.linalg.sqrtm(sa @ sb)).real
    assert abs(got - want) < 1e-9 * max(1.0, abs(want))
    assert abs(ev.frechet_classifier_distance_from_activations(a, a)) < 1e-9
    # commuting (diagonal) covariances: sum_i (s_i - t_i)^2 + |dm|^2, with the sample statistics themselves
    d1, d2 = rng.standard_normal((500, 3)) * [1, 2, 3], rng.standard_normal((500, 3)) * [2, 2, 1] + [1, 0, 0]
    # float32 inputs are promoted to float64 like tfgan does
    f32 = ev.frechet_classifier_distance_from_activations(d1.astype(np.float32), d2.astype(np.float32))
    f64 = ev.frechet_classifier_distance_from_activations(d1, d2)
    assert abs(f32 - f64) < 1e-4 * f64
    # rank-deficient covariance (fewer samples than features) stays finite and non-negative
    assert ev.frechet_classifier_distance_from_activations(rng.standard_normal((5, 16)), rng.standard_normal((7, 16))) >= -1e-9


def test_inception_score_closed_forms():
    n, k = 64, 10
    assert abs(ev.classifier_score_from_logits(np.zeros((n, k))) - 1.0) < 1e-12  # uninformative classifier
    onehot = np.full((n, k), -1e4)
    onehot[np.arange(n), np.arange(n) % k] = 0.0  # confident and uniform over classes -> IS = number of classes hit equally
    hits = np.bincount(np.arange(n) % k, minlength=k) / n
    want = np.exp(-(hits * np.log(hits)).sum())
    assert abs(ev.classifier_score_from_logits(onehot) - want) < 1e-9


def test_check_fid_report(tmp_path):
    cfg = ConfigDict()
    cfg.data = ConfigDict(); cfg.data.dataset = 'CIFAR10'; cfg.data.image_size = 32; cfg.data.num_channels = 3
    cfg.eval = ConfigDict(); cfg.eval.num_samples = 10; cfg.eval.batch_size = 4
    rng = np.random.default_rng(1)
    res = tmp_path / "res"; res.mkdir()
    (tmp_path / "assets" / "stats").mkdir(parents=True)
    data_pool = rng.standard_normal((50, 8))
    np.savez(tmp_path / "assets" / "stats" / "cifar10_stats.npz", pool_3=data_pool)
    with pytest.raises(Exception, match="inception"):
        run_lib.check_fid(cfg, str(res), stats_root=str(tmp_path))
    feats = {}
    for r in range(3):  # 10 // 4 + 1 rounds
        np.savez_compressed(res / ("samples_%d.npz" % r), samples=rng.integers(0, 255, (4, 32, 32, 3), dtype=np.uint8), nfe_cnt=50)

    def fake_inception(images):
        f = {"pool_3": rng.standard_normal((images.shape[0], 8)), "logits": rng.standard_normal((images.shape[0], 5))}
        feats.setdefault("pool", []).append(f["pool_3"])
        return f

    rep = run_lib.check_fid(cfg, str(res), inception_fn=fake_inception, stats_root=str(tmp_path))
    pools = np.concatenate(feats["pool"])[:10]
    assert abs(rep["fid"] - ev.frechet_classifier_distance_from_activations(data_pool, pools)) < 1e-9 and rep["nfe"] == 50
    saved = np.load(res / "report.npz")
    assert set(saved.files) == {"IS", "fid", "nfe"} and abs(float(saved["fid"]) - rep["fid"]) < 1e-12
    # second call reuses statistics_<r>.npz without the network
    rep2 = run_lib.check_fid(cfg, str(res), stats_root=str(tmp_path))
    assert abs(rep2["fid"] - rep["fid"]) < 1e-12
