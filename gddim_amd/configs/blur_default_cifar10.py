"""Field names / defaults of blur_jax/configs/default_cifar10_config.py:4-80."""
from ..config_dict import ConfigDict


def get_default_configs():
    config = ConfigDict()
    config.training = training = ConfigDict()
    training.batch_size = 128
    training.n_iters = 1300001
    training.snapshot_freq = 50000
    training.log_freq = 50
    training.eval_freq = 100
    training.snapshot_sampling = True
    training.likelihood_weighting = False
    training.continuous = True
    training.n_jitted_steps = 5
    training.reduce_mean = True
    training.snapshot_freq_for_preemption = 50000
    training.snapshot_sampling_batch = 100

    config.sampling = sampling = ConfigDict()
    sampling.method = 'order0'
    sampling.nfe = 50
    sampling.ts_order = 2
    sampling.t0 = 1e-5

    config.eval = evaluate = ConfigDict()
    evaluate.begin_ckpt = 9
    evaluate.end_ckpt = 26
    evaluate.batch_size = 1024
    evaluate.enable_sampling = False
    evaluate.num_samples = 50000
    evaluate.enable_loss = True
    evaluate.enable_bpd = False
    evaluate.bpd_dataset = 'test'

    config.data = data = ConfigDict()
    data.dataset = 'CIFAR10'
    data.image_size = 32
    data.random_flip = True
    data.centered = False
    data.uniform_dequantization = False
    data.num_channels = 3
    data.is_partial = False

    config.model = model = ConfigDict()
    model.sigma_min = 0.01
    model.sigma_max = 50
    model.num_scales = 1000
    model.beta_min = 0.1
    model.beta_max = 20.
    model.dropout = 0.1
    model.embedding_type = 'fourier'
    model.sigma_blur_max = 10.0

    config.optim = optim = ConfigDict()
    optim.weight_decay = 0
    optim.optimizer = 'Adam'
    optim.lr = 2e-4
    optim.beta1 = 0.9
    optim.eps = 1e-8
    optim.warmup = 5000
    optim.grad_clip = 1.

    config.seed = 42
    return config
