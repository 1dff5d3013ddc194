"""Default field values shared by the reference's two `default_cifar10_config.py` files, as data.

The reference builds its configs by assigning one field per line to ml_collections.ConfigDict objects; here the same field names and
values live in nested dict literals (`COMMON`, then the CLD- and blur-specific additions) and `make()` turns them into a ConfigDict.
Only the names and values matter to the engine, which reads a handful of them."""
import copy

from ..config_dict import ConfigDict

COMMON = {
    'training': dict(batch_size=128, n_iters=1300001, snapshot_freq=50000, log_freq=50, eval_freq=100, snapshot_sampling=True,
                     likelihood_weighting=False, continuous=True, n_jitted_steps=5, reduce_mean=True,
                     snapshot_freq_for_preemption=50000, snapshot_sampling_batch=100),
    'sampling': dict(ts_order=2),
    'eval': dict(begin_ckpt=9, end_ckpt=26, batch_size=1024, enable_sampling=False, num_samples=50000, enable_loss=True,
                 enable_bpd=False, bpd_dataset='test'),
    'data': dict(dataset='CIFAR10', image_size=32, random_flip=True, centered=False, uniform_dequantization=False, num_channels=3),
    'model': dict(sigma_min=0.01, sigma_max=50, num_scales=1000, beta_min=0.1, beta_max=20., dropout=0.1, embedding_type='fourier'),
    'optim': dict(weight_decay=0, optimizer='Adam', lr=2e-4, beta1=0.9, eps=1e-8, warmup=5000, grad_clip=1.),
    'seed': 42,
}

# cld_jax/configs/default_cifar10_config.py: the CLD SDE constants, its R(t) scan switches and the sampler menu
CLD = {
    'training': dict(ema_update_freq=1e9),
    'sampling': dict(n_steps_each=1, noise_removal=True, probability_flow=False, snr=0.16, method='deis', nfe=20, is_em=False,
                     deis_order=1, noise_nfe_ratio=0.3, img_t_ratio=0.3, atol=1e-5, rtol=1e-5, ode_method='RK45', lambda_coef=1.0,
                     sdeis_use_order0=True),
    'model': dict(m_inv=4.0, beta_0=4.0, beta_1=0.0, vv_gamma=0.04, mixed_score=False, is_R_rk=False, R_dt=1e-5, used_cache=True,
                  x64=False),
}

# blur_jax/configs/default_cifar10_config.py
BLUR = {
    'sampling': dict(method='order0', nfe=50, t0=1e-5),
    'data': dict(is_partial=False),
    'model': dict(sigma_blur_max=10.0),
}

# the NCSN++ "deep" architecture of accr_dcifar10 / deep / ddpm_deep (model.* fields)
NCSNPP_DEEP = dict(name='ncsnpp', scale_by_sigma=False, ema_rate=0.9999, normalization='GroupNorm', nonlinearity='swish', nf=128,
                   ch_mult=(1, 2, 2, 2), num_res_blocks=8, attn_resolutions=(16,), resamp_with_conv=True, conditional=True, fir=True,
                   fir_kernel=[1, 3, 3, 1], skip_rescale=True, resblock_type='biggan', progressive='none',
                   progressive_input='residual', progressive_combine='sum', attention_type='ddpm', init_scale=0.,
                   embedding_type='fourier', fourier_scale=16, conv_size=3)


def make(*layers):
    """Merge section dicts left to right and return a ConfigDict tree."""
    merged = {}
    for layer in layers:
        for section, value in layer.items():
            if isinstance(value, dict):
                merged.setdefault(section, {}).update(copy.deepcopy(value))
            else:
                merged[section] = value
    config = ConfigDict()
    for section, value in merged.items():
        if isinstance(value, dict):
            sub = ConfigDict()
            for k, v in value.items():
                setattr(sub, k, v)
            setattr(config, section, sub)
        else:
            setattr(config, section, value)
    return config
