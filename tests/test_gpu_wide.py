"""GPU parity of the wide-hidden kernels (256 < H <= 1024: BASELINE.json configs[3] "wider hidden size" and
configs[4] "wide-hidden BRNN"; timediff_amd/csrc/rnn_wide.hip) against the CPU oracle.

The reference holds no vectors at these widths and a golden fixture would carry > 10 MB of weights, so the
oracle here is the same C restatement that tests/test_oracle_golden.py pins against the reference at H <= 92
(it is generic in H); weights are torch-default-initialised from a seed.  Tolerances as in test_gpu_parity.py.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from test_gpu_parity import BF16_EMUL_ATOL, _rel, cu, den_kernels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def random_state(H, C, layers, dirs, gru, seed):
    g = torch.Generator().manual_seed(seed)
    k = 1.0 / np.sqrt(H)
    G = 3 if gru else 4
    sd = {}
    for l in range(layers):
        for sfx in ("", "_reverse")[:dirs]:
            In = C if l == 0 else dirs * H
            sd[f"rnn.rnn.weight_ih_l{l}{sfx}"] = (torch.rand(G * H, In, generator=g) * 2 - 1) * k
            sd[f"rnn.rnn.weight_hh_l{l}{sfx}"] = (torch.rand(G * H, H, generator=g) * 2 - 1) * k
            sd[f"rnn.rnn.bias_ih_l{l}{sfx}"] = (torch.rand(G * H, generator=g) * 2 - 1) * k
            sd[f"rnn.rnn.bias_hh_l{l}{sfx}"] = (torch.rand(G * H, generator=g) * 2 - 1) * k
    DH = dirs * H
    sd["time_emb.1.weight"] = torch.randn(256, 64, generator=g) / 8; sd["time_emb.1.bias"] = torch.zeros(256)
    sd["time_emb.3.weight"] = torch.randn(256, 256, generator=g) / 16; sd["time_emb.3.bias"] = torch.zeros(256)
    sd["rnn.time_mlp.1.weight"] = torch.randn(2 * DH, 256, generator=g) / 16; sd["rnn.time_mlp.1.bias"] = torch.zeros(2 * DH)
    sd["rnn.norm.weight"] = 1 + 0.1 * torch.randn(DH, generator=g); sd["rnn.norm.bias"] = 0.1 * torch.randn(DH, generator=g)
    sd["fc.weight"] = torch.randn(C, DH, generator=g) / np.sqrt(DH); sd["fc.bias"] = 0.1 * torch.randn(C, generator=g)
    return {k_: v.numpy() for k_, v in sd.items()}


def _inputs(B, C, L, n_num, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, C, L)).astype(np.float32)
    x[:, n_num:] = np.where(rng.random((B, C - n_num, L)) > 0.5, 0.0, -69.07755279).astype(np.float32)
    return x, rng.integers(0, 1000, B), rng


# (H, layers, dirs, gru, C, n_num, B, L): HiRID-shaped (C_io = 9) and eICU-shaped (14) channel maps
WIDE = [
    (512, 3, 2, 0, 14, 4, 37, 9),     # 16 unit blocks, ragged batch
    (320, 2, 2, 1, 9, 7, 33, 7),      # GRU, 10 blocks padded to 12
    (1024, 2, 2, 0, 9, 7, 32, 5),     # 32 blocks: 128 KiB of LDS for h
    (768, 2, 1, 0, 23, 7, 5, 6),      # unidirectional, 24 blocks, two layer-0 input k-steps
    (400, 2, 1, 1, 14, 4, 64, 4),     # GRU padded 13 -> 16 blocks
]


@pytest.mark.parametrize("H,layers,dirs,gru,C,n_num,B,L", WIDE)
def test_wide_denoiser_forward(H, layers, dirs, gru, C, n_num, B, L, dev):
    sd = random_state(H, C, layers, dirs, gru, seed=H + layers)
    den = den_kernels(sd, dev)
    x, t, _ = _inputs(B, C, L, n_num, seed=H)
    y, hidden = den.forward(cu(x, dev), cu(t, dev), want_hidden=True)
    y, hidden = y.cpu().numpy(), hidden.cpu().numpy()
    y16, h16 = orc.Denoiser(sd, bf16=True).forward(x, t, want_hidden=True)
    assert np.isfinite(y).all()
    assert np.abs(hidden - h16).max() < BF16_EMUL_ATOL, np.abs(hidden - h16).max()
    assert np.abs(y - y16).max() < 2e-2, np.abs(y - y16).max()
    y32 = orc.Denoiser(sd).forward(x, t)
    assert np.abs(y - y32).max() < 6e-2, np.abs(y - y32).max()
    # one shared FiLM row (the sampler's call: every sequence at the same step)
    t1 = np.full(B, 321)
    y1 = den.forward(cu(x, dev), film=den.time_film(cu(t1[:1], dev))).cpu().numpy()
    assert np.abs(y1 - orc.Denoiser(sd, bf16=True).forward(x, t1)).max() < 2e-2


def test_wide_sample_chain_matches_stepwise(dev):
    """HiRID-shaped wide sampler (C_io = 9, H = 512): the captured-graph chain equals the eager step-by-step chain
    bit for bit, and the decoded samples are in range."""
    from timediff_amd.ops import DiffusionKernels, SamplerKernels
    H, L, T, B = 512, 12, 8, 40
    num, cat, K = [0, 1, 2, 3, 4, 5, 6], [7], [2]
    sd = random_state(H, 9, 3, 2, 0, seed=11)
    den = den_kernels(sd, dev)
    tables = {k: torch.from_numpy(v) for k, v in orc.schedule(T, "cosine").items()}
    diff = DiffusionKernels(num, cat, K, L, T, tables, 8, dev)
    sampler = SamplerKernels(den, diff)
    a = sampler.sample(B, seed=5, patient0=100).cpu().numpy()
    b = sampler.sample(B, seed=5, patient0=100, use_graph=False).cpu().numpy()
    np.testing.assert_array_equal(a, b)
    assert a.shape == (B, 8, L) and np.isfinite(a).all()
    assert ((a[:, cat] == 0) | (a[:, cat] == 1)).all() and a[:, num].min() >= 0 and a[:, num].max() <= 1


# (H, layers, dirs, gru, C, n_num, B, L)
WIDE_TRAIN = [
    (512, 2, 2, 0, 14, 4, 37, 8),     # bidirectional, head backward with parameters in LDS
    (512, 3, 2, 1, 9, 7, 70, 6),      # GRU at 16 unit blocks: the three-term resident-weight d x GEMM, three batch blocks
    (320, 2, 2, 1, 9, 7, 33, 6),      # GRU, padded unit blocks
    (1024, 2, 2, 0, 9, 7, 20, 5),     # D*H = 2048, 4 unit blocks per sweep wave
    (768, 2, 1, 0, 23, 7, 5, 6),      # c_out > 16
    (1024, 1, 2, 0, 23, 7, 4, 3),     # c_out > 16 with D*H = 2048: the head backward reads LayerNorm / fc weights from L2
]


@pytest.mark.parametrize("H,layers,dirs,gru,C,n_num,B,L", WIDE_TRAIN)
def test_wide_denoiser_backward(H, layers, dirs, gru, C, n_num, B, L, dev):
    """Wide training path: the saving forward equals the inference forward bit for bit; parameter gradients against
    the fp32 oracle's BPTT within the bf16 bound (3 % of each tensor's largest entry)."""
    sd = random_state(H, C, layers, dirs, gru, seed=H + 7)
    den = den_kernels(sd, dev)
    x, t, rng = _inputs(B, C, L, n_num, seed=H + 1)
    gy = (rng.standard_normal((B, C, L)) / (B * L)).astype(np.float32)
    film = den.time_film(cu(t, dev))
    y_inf = den.forward(cu(x, dev), film=film).cpu().numpy()
    y_tr = den.forward_train(cu(x, dev), film).cpu().numpy()
    np.testing.assert_array_equal(y_tr, y_inf)
    g_layer, g_misc, g_film = den.backward(cu(gy, dev), film)
    _, _, ref = orc.Denoiser(sd).backward(x, t, gy)
    for key, got in list(zip(den._lkeys, g_layer)) + list(g_misc.items()):
        assert _rel(got.cpu().numpy(), ref[key]) < 3e-2, (key, _rel(got.cpu().numpy(), ref[key]))
    np.testing.assert_allclose(g_film.sum(0).cpu().numpy(), ref["rnn.time_mlp.1.bias"], rtol=3e-2,
                               atol=3e-2 * np.abs(ref["rnn.time_mlp.1.bias"]).max())


@pytest.fixture
def step_launch():
    from timediff_amd import _lib

    def set_mode(mode):
        _lib.check(_lib.lib().td_set_step_launch(mode))
    yield set_mode
    _lib.check(_lib.lib().td_set_step_launch(_lib.STEP_LAUNCH_AUTO))
    _lib.check(_lib.lib().td_set_layer_kernel(_lib.LAYER_KERNEL_AUTO))


# (H, layers, dirs, gru, C, n_num, B, L)
STEP_LAUNCH = [
    (512, 3, 2, 0, 14, 4, 70, 6),       # 16 unit blocks, three batch blocks, the last ragged
    (320, 2, 2, 1, 9, 7, 33, 5),        # GRU, 10 blocks padded to 12
    (768, 2, 1, 0, 23, 7, 40, 4),       # unidirectional, two layer-0 input k-steps
    (1024, 2, 2, 0, 9, 7, 1040, 3),     # two batch blocks per wave (more than 512 tasks), an odd block count: the last twice
    (256, 2, 2, 0, 14, 4, 40, 6),       # forced below the widths AUTO takes: 8 blocks, bias gradients through the reduction
    (128, 2, 1, 1, 9, 7, 33, 5),        # ... 4 blocks, GRU
]


@pytest.mark.parametrize("H,layers,dirs,gru,C,n_num,B,L", STEP_LAUNCH)
def test_step_launch_equals_persistent_kernels(H, layers, dirs, gru, C, n_num, B, L, dev, step_launch):
    """The unit-split family as one launch per time step (rnn_wide.hip: td_rnn_step_wide_kernel, rnn_train.hip:
    td_bptt_step_kernel; AUTO takes it above H = 256 at small batches) against the persistent kernels of the same family:
    the same MFMA order and gate arithmetic, so outputs, saved activations' consequences and every gradient are bit-identical
    (bias gradients of widths up to 256 aside: the persistent sweep sums d a in fp32 registers, the per-step form reduces
    the bf16 copy the GEMMs read, as every wide layer does)."""
    from timediff_amd import _lib
    sd = random_state(H, C, layers, dirs, gru, seed=H + B)
    den = den_kernels(sd, dev)
    x, t, rng = _inputs(B, C, L, n_num, seed=H + 3)
    gy = (rng.standard_normal((B, C, L)) / (B * L)).astype(np.float32)
    if H <= 256:
        _lib.check(_lib.lib().td_set_layer_kernel(_lib.LAYER_KERNEL_UNITS))
    film = den.time_film(cu(t, dev))
    out = {}
    for mode in (_lib.STEP_LAUNCH_OFF, _lib.STEP_LAUNCH_ON):
        step_launch(mode)
        y, hidden = den.forward(cu(x, dev), cu(t, dev), want_hidden=True)
        y_tr = den.forward_train(cu(x, dev), film)
        g_layer, g_misc, g_film = den.backward(cu(gy, dev), film)
        out[mode] = dict(y=y, hidden=hidden, y_tr=y_tr, g_film=g_film, **{k: v.clone() for k, v in zip(den._lkeys, g_layer)},
                         **{"misc." + k: v.clone() for k, v in g_misc.items()})
    a, b = out[_lib.STEP_LAUNCH_OFF], out[_lib.STEP_LAUNCH_ON]
    assert torch.isfinite(b["y"]).all()
    for key in a:
        if H <= 256 and "bias_" in key:
            assert _rel(b[key].cpu().numpy(), a[key].cpu().numpy()) < 1e-2, key
        elif key.startswith("misc."):      # the head's parameter gradients are sums of atomics: equal up to their order
            assert _rel(b[key].cpu().numpy(), a[key].cpu().numpy()) < 1e-5, key
        else:
            assert torch.equal(a[key], b[key]), key
    step_launch(_lib.STEP_LAUNCH_AUTO)
    want = _lib.STEP_LAUNCH_ON if H > 256 and -(-B // 32) * dirs <= 128 else _lib.STEP_LAUNCH_OFF
    import ctypes as C
    assert _lib.lib().td_rnn_step_launch_choice(C.byref(den.cfg), B, L) == (1 if want == _lib.STEP_LAUNCH_ON else 0)
    np.testing.assert_array_equal(den.forward(cu(x, dev), cu(t, dev)).cpu().numpy(), out[want]["y"].cpu().numpy())
