"""ORACLE -- CPU restatement of the reference's gDDIM sampling hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package, and only as the checker.
The product (``gddim_amd``) never imports it and fails loudly when its HIP library is
missing.

PARITY UNPINNED: the reference (qsh-zh/gDDIM) is pure Python on jax==0.2.8 / flax==0.3.1 /
ml_collections / jammy (``/root/reference/Dockerfile:1-15``); none is installed here, there
is no network, and the reference's own tests (``cld_jax/tests/*``) contain no assertion,
fixture or golden vector.  So this oracle cannot be checked against reference outputs.
It is instead written line-by-line from the cited reference sources and pinned by analytic
known-answer tests (``tests/test_oracle_*.py``): matrix exponential vs scipy ``expm``,
``R R^T`` vs the Lyapunov ODE, DEIS coefficients vs adaptive quadrature and the Lagrange
row-sum identity, DCT vs ``scipy.fft.dctn``, FIR resampling vs ``scipy.signal.upfirdn``,
conv/GroupNorm/softmax vs independent numpy loops, and the Dirac-data closed-form sampler
trajectory.  Third-party arithmetic the reference delegates to (flax ``nn.Conv`` /
``nn.GroupNorm(epsilon=1e-6)`` / ``nn.Dense``, ``jax.nn.softmax``, ``lax.conv_general_dilated``)
is restated from its published semantics.

PINNING IT IS ONE COMMAND AWAY: ``tools/make_reference_fixtures.py`` (run once inside the reference's own image) imports ``cld_jax`` /
``blur_jax``, and writes the reference's coefficient tables, R(t) anchors, a tiny-NCSN++ ``eps_fn`` output with its flax checkpoint,
a 5-step ``get_deis_sampler`` run with injected noise and a blur ``order0`` run to ``tests/golden/ref_*``;
``tests/test_reference_fixtures.py`` then checks this oracle (CPU) and the HIP path (GPU) against them and skips while they are absent.

Modules: ``cld`` (CLD SDE algebra, DEIS tables, deis / order0 samplers), ``cld_ext`` (LSDE, MLCLD, LambdaSDE and the ldeis /
mldeis / sdeis / em / sscs / ode samplers, mixed score), ``ncsnpp`` (the U-Net incl. the fir=False / positional-embedding
configurations), ``blur`` (blur SDE, DCT, order0 sampler, scalar DEIS; plus an independent float64 formulation of the package's
per-frequency multistep extension, marked as not being reference code), ``rng`` (restatement of the ENGINE's Philox / Box-Muller
noise source -- jax's threefry stream cannot be reproduced -- pinned by the Random123 known-answer vectors), ``rscan.c`` (R(t) scan
in float32 / float64).
"""
