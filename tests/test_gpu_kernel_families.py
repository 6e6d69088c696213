"""GPU parity of BOTH recurrent-layer kernel families at hidden sizes <= 256, each pinned explicitly.

csrc/rnn.hip (td_use_wide) sends a launch either to the batch-block kernel of rnn_layer.h (a wave owns 32 sequences
and all units: the kernel bench.py times, taken above ~8192 sequences) or to the unit-split kernel of rnn_wide.hip
(small batches).  The golden-vector tests run at small batches, i.e. through the unit-split kernel wherever it
exists, so this file
  * forces the batch-block family with td_set_layer_kernel and walks EVERY instantiation rnn_launch.h lists
    (hidden blocks 1, 2, 3, 4, 6, 8 x LSTM / GRU x inference / saving; layer-0 input k-steps 1 and 2; one and two
    directions) against the oracle: forward (bf16-emulating oracle), saving forward == inference forward bit for bit,
    parameter gradients (fp32 oracle);
  * runs the AUTOMATIC dispatch just above each switch point (H = 256 / 128 above 8192 sequences, H = 192 / 92 / 40
    above 1024) and asserts through td_rnn_layer_kernel_choice that the batch-block kernel ran: the configuration
    bench.py measures (H = 256, 3 layers, bidirectional, B > 8192) is checked at full width with a short sequence.
    Sequences are independent, so the oracle is evaluated on a subset of them (first tiles, strided middle, the
    ragged tail); gradients are checked with grad_y non-zero on that subset only (every other sequence still
    flows through every kernel and must contribute exactly nothing).
Tolerances as in test_gpu_parity.py (reference semantics: blocks.py:282-297).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from test_gpu_parity import BF16_EMUL_ATOL, _rel, cu, den_kernels
from test_gpu_wide import _inputs, random_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture()
def knob():
    """td_set_layer_kernel, restored to AUTO afterwards."""
    from timediff_amd import _lib
    L = _lib.lib()

    def set_mode(mode):
        _lib.check(L.td_set_layer_kernel(mode))
    yield set_mode
    L.td_set_layer_kernel(_lib.LAYER_KERNEL_AUTO)


@pytest.fixture()
def fusion():
    """td_set_head_fusion, restored to AUTO afterwards."""
    from timediff_amd import _lib
    L = _lib.lib()

    def set_mode(mode):
        _lib.check(L.td_set_head_fusion(mode))
    yield set_mode
    L.td_set_head_fusion(_lib.HEAD_FUSION_AUTO)


def _choice(den, B, L=6):
    from timediff_amd import _lib
    return _lib.lib().td_rnn_layer_kernel_choice(C.byref(den.cfg), B, L)


def _check_forward_and_grads(den, sd, x, t, gy, rows, dev, tol_y=2e-2):
    """forward vs the bf16-emulating oracle on `rows`, saving forward == inference forward on ALL rows,
    parameter gradients vs the fp32 oracle (grad_y is zero outside `rows`).  Hidden state: 1e-2 (one bf16 step of an
    O(1) value); the projected output sums D*H of them: 2e-2, as in test_gpu_wide.py."""
    film = den.time_film(cu(t, dev))
    y, hidden = den.forward(cu(x, dev), film=film, want_hidden=True)
    y, hidden = y.cpu().numpy(), hidden.cpu().numpy()
    assert np.isfinite(y).all()
    y16, h16 = orc.Denoiser(sd, bf16=True).forward(x[rows], t[rows], want_hidden=True)
    assert np.abs(hidden[rows] - h16).max() < BF16_EMUL_ATOL, np.abs(hidden[rows] - h16).max()
    assert np.abs(y[rows] - y16).max() < tol_y, np.abs(y[rows] - y16).max()
    y_tr = den.forward_train(cu(x, dev), film).cpu().numpy()
    np.testing.assert_array_equal(y_tr, y)
    g_layer, g_misc, g_film = den.backward(cu(gy, dev), film)
    _, _, ref = orc.Denoiser(sd).backward(x[rows], t[rows], gy[rows])
    for key, got in list(zip(den._lkeys, g_layer)) + list(g_misc.items()):
        got = got.cpu().numpy()
        assert np.isfinite(got).all(), key
        assert _rel(got, ref[key]) < 3e-2, (key, _rel(got, ref[key]))
    g_film = g_film.cpu().numpy()
    off = np.ones(len(x), bool)
    off[rows] = False
    assert not g_film[off].any(), "a sequence with zero grad_y produced a FiLM gradient"
    np.testing.assert_allclose(g_film.sum(0), ref["rnn.time_mlp.1.bias"], rtol=3e-2,
                               atol=3e-2 * np.abs(ref["rnn.time_mlp.1.bias"]).max())


# hidden sizes per unit-block count: one exact multiple of 32 and one padded width each
NU_WIDTHS = {1: (32, 20), 2: (64, 40), 3: (96, 92), 4: (128, 100), 6: (192, 170), 8: (256, 250)}


@pytest.mark.parametrize("gru", [0, 1])
@pytest.mark.parametrize("nu", [1, 2, 3, 4, 6, 8])
@pytest.mark.parametrize("shape", ["bi_c14", "uni_c23"])
def test_batch_block_family_forced(nu, gru, shape, dev, knob):
    """Every template instantiation of td_rnn_layer_kernel: <NU, 1|2, L0>, <NU, 2NU|4NU, upper> x GRU x SAVE."""
    from timediff_amd import _lib
    knob(_lib.LAYER_KERNEL_BATCH)
    dirs, Cc, n_num = (2, 14, 4) if shape == "bi_c14" else (1, 23, 7)
    H = NU_WIDTHS[nu][0 if shape == "bi_c14" else 1]
    B, L, layers = 150, 5, 3        # 5 batch blocks: the second workgroup has one live wave and three clamped ones
    sd = random_state(H, Cc, layers, dirs, gru, seed=100 * nu + 10 * gru + dirs)
    den = den_kernels(sd, dev)
    assert _choice(den, B) == _lib.LAYER_KERNEL_BATCH
    x, t, rng = _inputs(B, Cc, L, n_num, seed=nu)
    gy = (rng.standard_normal((B, Cc, L)) / (B * L)).astype(np.float32)
    _check_forward_and_grads(den, sd, x, t, gy, np.arange(B), dev)


@pytest.mark.parametrize("gru", [0, 1])
@pytest.mark.parametrize("nu", [1, 2, 3])
@pytest.mark.parametrize("shape", ["bi_c14", "uni_c23"])
def test_resident_family_forced(nu, gru, shape, dev, knob):
    """Every instantiation of td_rnn_resident_kernel (widths of up to three unit blocks: the whole layer in one workgroup,
    weights in registers, h through LDS): <NU, 2|4, L0>, <NU, 2NU|4NU, upper> x GRU x SAVE, exact and padded widths, with
    a ragged last batch block."""
    from timediff_amd import _lib
    knob(_lib.LAYER_KERNEL_CLUSTER)
    dirs, Cc, n_num = (2, 14, 4) if shape == "bi_c14" else (1, 23, 7)
    H = NU_WIDTHS[nu][0 if shape == "bi_c14" else 1]
    B, L, layers = 150, 7, 3
    sd = random_state(H, Cc, layers, dirs, gru, seed=200 * nu + 10 * gru + dirs)
    den = den_kernels(sd, dev)
    assert _choice(den, B, L) == _lib.LAYER_KERNEL_CLUSTER
    x, t, rng = _inputs(B, Cc, L, n_num, seed=nu + 50)
    gy = (rng.standard_normal((B, Cc, L)) / (B * L)).astype(np.float32)
    _check_forward_and_grads(den, sd, x, t, gy, np.arange(B), dev)


@pytest.mark.parametrize("L", [1, 2, 3])
def test_resident_kernels_shortest_sequences(L, dev):
    """One, two and three steps (no hand-over at all / one / two): the loops' first and last iterations are special."""
    from timediff_amd import _lib
    H, Cc, n_num, B = 40, 14, 4, 70
    sd = random_state(H, Cc, 2, 2, 0, seed=L)
    den = den_kernels(sd, dev)
    assert _choice(den, B, L) == _lib.LAYER_KERNEL_CLUSTER
    x, t, rng = _inputs(B, Cc, L, n_num, seed=L + 3)
    gy = (rng.standard_normal((B, Cc, L)) / (B * L)).astype(np.float32)
    _check_forward_and_grads(den, sd, x, t, gy, np.arange(B), dev)


@pytest.mark.parametrize("H,Cc,n_num,L,gru", [(92, 23, 7, 24, 0), (36, 9, 7, 100, 0), (36, 9, 7, 100, 1)])
def test_resident_kernels_at_the_reference_shapes(H, Cc, n_num, L, gru, dev):
    """The reference's default widths and lengths (etdiff_train.py:24-29, :239: MIMIC-III H = 92, L = 24; HiRID H = 36,
    L = 100) at batch 256 under AUTO: the one-workgroup kernels, forward and BPTT over the whole sequence length (99
    hand-overs of h / partial sums through LDS), oracle on a subset of the sequences."""
    from timediff_amd import _lib
    B = 256
    sd = random_state(H, Cc, 3, 2, gru, seed=H + L)
    den = den_kernels(sd, dev)
    assert _choice(den, B, L) == _lib.LAYER_KERNEL_CLUSTER
    x, t, rng = _inputs(B, Cc, L, n_num, seed=L)
    rows = np.unique(np.concatenate([np.arange(0, 34), np.arange(0, B, 29), np.arange(B - 20, B)]))
    gy = np.zeros((B, Cc, L), np.float32)
    gy[rows] = (rng.standard_normal((len(rows), Cc, L)) / (len(rows) * L)).astype(np.float32)
    _check_forward_and_grads(den, sd, x, t, gy, rows, dev)


# (H, gru, dirs, B): just above the automatic switch to the batch-block kernel
AUTO_CASES = [
    (256, 0, 2, 8192 + 40),     # bench.py's configuration (BASELINE.json configs[1]/[2]): <8, 1, L0> and <8, 32, upper>
    (256, 1, 2, 8192 + 40),
    (128, 0, 2, 8192 + 40),
    (256, 0, 1, 16384 + 40),
    (192, 0, 2, 1024 + 40),
    (92, 0, 2, 16384 + 40),     # MIMIC-III width (configs[0]) at the large-batch sampler's side of the switch
    (40, 1, 2, 16384 + 40),     # HiRID-like width
]


@pytest.mark.parametrize("H,gru,dirs,B", AUTO_CASES)
def test_auto_dispatch_above_switch(H, gru, dirs, B, dev):
    from timediff_amd import _lib
    L, layers, Cc, n_num = 6, 3, 14, 4
    sd = random_state(H, Cc, layers, dirs, gru, seed=H + gru)
    den = den_kernels(sd, dev)
    assert _choice(den, B) == _lib.LAYER_KERNEL_BATCH, "this batch should run the batch-block kernel"
    # small batches: the cluster family at the widths it is built for (8 unit blocks over 8 CUs; up to 3 in one workgroup)
    small = _lib.LAYER_KERNEL_CLUSTER if H > 192 or H <= 96 else _lib.LAYER_KERNEL_UNITS
    assert _choice(den, 64) == small
    x, t, rng = _inputs(B, Cc, L, n_num, seed=H)
    rows = np.unique(np.concatenate([np.arange(0, 40), np.arange(0, B, 97), np.arange(B - 48, B)]))
    gy = np.zeros((B, Cc, L), np.float32)
    gy[rows] = (rng.standard_normal((len(rows), Cc, L)) / (len(rows) * L)).astype(np.float32)
    _check_forward_and_grads(den, sd, x, t, gy, rows, dev)


def test_families_agree_on_the_same_batch(dev, knob):
    """The two families differ in summation order only: same inputs, outputs within the bf16 bound of each other,
    and each one is deterministic run to run."""
    from timediff_amd import _lib
    H, Cc, B, L = 256, 14, 300, 7
    sd = random_state(H, Cc, 3, 2, 0, seed=3)
    den = den_kernels(sd, dev)
    x, t, _ = _inputs(B, Cc, L, 4, seed=9)
    out = {}
    for mode in (_lib.LAYER_KERNEL_BATCH, _lib.LAYER_KERNEL_UNITS):
        knob(mode)
        assert _choice(den, B) == mode
        a = den.forward(cu(x, dev), cu(t, dev)).cpu().numpy()
        b = den.forward(cu(x, dev), cu(t, dev)).cpu().numpy()
        np.testing.assert_array_equal(a, b)
        out[mode] = a
    assert np.abs(out[_lib.LAYER_KERNEL_BATCH] - out[_lib.LAYER_KERNEL_UNITS]).max() < BF16_EMUL_ATOL


def test_large_batch_chain_graph_equals_stepwise(dev, fusion):
    """The sampler path bench.py runs (diffusion state -> layer-0 fragments -> batch-block layers with the head fused
    into the top layer -> reverse update, replayed as a hipGraph) above the switch: the graph chain equals the eager
    chain bit for bit, and equals the tensor-level composition (build_input -> forward -> reverse_step_) run with the
    same fused head; with the fusion off the chain equals the composition through td_head_kernel, whose pieces the
    other tests pin to the oracle; the two chains agree within the rounding of the projection."""
    from timediff_amd import _lib
    from timediff_amd.ops import DiffusionKernels, SamplerKernels
    H, L, T, B = 256, 6, 3, 8192 + 40
    num, cat, K = [0, 2, 4, 6], [1, 3, 5, 7, 8], [2] * 5
    sd = random_state(H, 14, 3, 2, 0, seed=21)
    den = den_kernels(sd, dev)
    assert _choice(den, B) == _lib.LAYER_KERNEL_BATCH
    tables = {k: torch.from_numpy(v) for k, v in orc.schedule(T, "cosine").items()}
    diff = DiffusionKernels(num, cat, K, L, T, tables, 9, dev)
    sampler = SamplerKernels(den, diff)
    film = den.time_film(torch.arange(T, device=dev))

    def composed():
        z, idx = diff.prior(B, seed=5, patient0=1000)
        for t in reversed(range(T)):
            mo = den.forward(diff.build_input(z, idx), film=film[t:t + 1])
            diff.reverse_step_(mo, z, idx, t, seed=5, patient0=1000)
        return diff.decode(z, idx)

    chains = {}
    for mode in (_lib.HEAD_FUSION_ON, _lib.HEAD_FUSION_OFF):      # ON: the chain fuses (as under AUTO) and so does forward
        fusion(mode)
        a = sampler.sample(B, seed=5, patient0=1000)
        b = sampler.sample(B, seed=5, patient0=1000, use_graph=False)
        assert torch.equal(a, b)
        assert torch.equal(a, composed())
        chains[mode] = a.cpu().numpy()
    fusion(_lib.HEAD_FUSION_AUTO)
    assert np.array_equal(sampler.sample(B, seed=5, patient0=1000).cpu().numpy(), chains[_lib.HEAD_FUSION_ON])
    on, off = chains[_lib.HEAD_FUSION_ON], chains[_lib.HEAD_FUSION_OFF]
    # a class that flips at a near-tie changes that sequence's later steps: the chains agree up to a few such sequences
    assert (np.abs(on[:, num] - off[:, num]) > 2e-3).mean() < 5e-3
    assert (on[:, cat] != off[:, cat]).mean() < 2e-3
    # and the denoiser output of the first step against the oracle on a subset of sequences, both ways
    z, idx = diff.prior(B, seed=5, patient0=1000)
    x_in = diff.build_input(z, idx)
    rows = np.unique(np.concatenate([np.arange(0, 33), np.arange(0, B, 211), np.arange(B - 40, B)]))
    y16 = orc.Denoiser(sd, bf16=True).forward(x_in.cpu().numpy()[rows], np.full(len(rows), T - 1))
    mo = {}
    for mode in (_lib.HEAD_FUSION_ON, _lib.HEAD_FUSION_OFF):
        fusion(mode)
        mo[mode] = den.forward(x_in, film=film[T - 1:T]).cpu().numpy()
        assert np.abs(mo[mode][rows] - y16).max() < 2e-2
    assert np.abs(mo[_lib.HEAD_FUSION_ON] - mo[_lib.HEAD_FUSION_OFF]).max() < 1e-3


# (H, L, B, num, cat, K): shapes whose reverse update also packs the next step's layer-0 fragments (L % 4 == 0, K <= 8)
REVERSE_PACKS = [
    (92, 24, 70, [0, 2, 4, 6, 8, 10, 12], [1, 3, 5, 7, 9, 11, 13, 14], [2] * 8),   # MIMIC-III: two k-tiles of input, ragged batch
    (36, 100, 33, [0, 1, 2, 3, 4, 5, 6], [7], [2]),                                 # HiRID: L not a multiple of the 32-position tile
    (64, 36, 256, [1, 2], [0, 3, 4], [3, 2, 5]),                                    # mixed class counts (the 8-class kernel), L = 32 + 4
    (48, 8, 5, [0, 1, 2], [], []),                                                  # continuous channels only
    (48, 8, 40, [], [0, 1], [2, 4]),                                                # discrete variables only
    # enough 32 x 32 tiles to fill the CUs: one thread updates all channels of its positions (the cases above: four waves share them)
    (92, 8, 8192 + 40, [0, 2, 4, 6, 8, 10, 12], [1, 3, 5, 7, 9, 11, 13, 14], [2] * 8),
    (36, 100, 2048 + 5, [0, 1, 2, 3, 4, 5, 6], [7], [2]),
    (64, 36, 4096 + 33, [1, 2], [0, 3, 4], [3, 2, 5]),
]


@pytest.mark.parametrize("H,L,B,num,cat,K", REVERSE_PACKS)
def test_chain_reverse_update_packs_next_input(H, L, B, num, cat, K, dev):
    """From the second step on the chain's layer-0 fragments are written by the reverse update's kernel (diffusion.hip,
    FRAGS form) instead of a packing kernel of their own: the chain equals the tensor-level composition
    (build_input -> forward -> reverse_step_, each step packing its own input) bit for bit, graph and eager, with
    Philox draws and with external ones."""
    from timediff_amd.ops import DiffusionKernels, SamplerKernels
    T = 5
    C_io, Cd = len(num) + sum(K), len(num) + len(cat)
    sd = random_state(H, C_io, 2, 2, 0, seed=H + L)
    den = den_kernels(sd, dev)
    tables = {k: torch.from_numpy(v) for k, v in orc.schedule(T, "cosine").items()}
    diff = DiffusionKernels(num, cat, K, L, T, tables, Cd, dev)
    sampler = SamplerKernels(den, diff)
    film = den.time_film(torch.arange(T, device=dev))
    g = torch.Generator(device="cpu").manual_seed(B)
    noise = torch.randn(T + 1, B, max(len(num), 1), L, generator=g)[:, :, :len(num)].contiguous().to(dev)
    uni = torch.rand(T + 1, B, max(sum(K), 1), L, generator=g)[:, :, :sum(K)].contiguous().to(dev)

    def composed(ext):
        nz = (lambda j: noise[j] if ext and len(num) else None)
        un = (lambda j: uni[j] if ext and len(cat) else None)
        z, idx = diff.prior(B, nz(0), un(0), seed=5, patient0=1000)
        for j, t in enumerate(reversed(range(T))):
            mo = den.forward(diff.build_input(z, idx), film=film[t:t + 1])
            diff.reverse_step_(mo, z, idx, t, nz(j + 1), un(j + 1), seed=5, patient0=1000)
        return diff.decode(z, idx)

    for ext in (False, True):
        kw = dict(noise=noise if len(num) else None, uniforms=uni if len(cat) else None) if ext else dict(seed=5, patient0=1000)
        a = sampler.sample(B, **kw)
        b = sampler.sample(B, use_graph=False, **kw)
        assert torch.equal(a, b)
        assert torch.equal(a, composed(ext))
        assert torch.isfinite(a).all()


# (H, gru, dirs, C, n_num, B): above each switch to the batch-block family
FUSED_HEAD = [
    (256, 1, 2, 14, 4, 8192 + 40),      # GRU
    (250, 0, 1, 23, 7, 16384 + 40),     # unidirectional, padded width, more than 16 output channels (wider partials)
    (128, 0, 2, 14, 4, 8192 + 40),      # 4 unit blocks
    (92, 1, 2, 23, 7, 1024 + 40),       # MIMIC-III width, 3 unit blocks
    (20, 0, 2, 9, 7, 300),              # one unit block: every head slot follows the block's gate math
]


@pytest.mark.parametrize("H,gru,dirs,Cc,n_num,B", FUSED_HEAD)
def test_fused_head_matches_head_kernel(H, gru, dirs, Cc, n_num, B, dev, fusion, knob):
    """td_rnn_layer_kernel<..., HEAD> + td_head_combine_kernel against td_head_kernel on the same layers (one FiLM row
    for the batch): both project the same bf16 hidden state with fp32-accurate weights, and against the oracle."""
    from timediff_amd import _lib
    L = 5
    sd = random_state(H, Cc, 2, dirs, gru, seed=H + B)
    den = den_kernels(sd, dev)
    if H <= 96:
        knob(_lib.LAYER_KERNEL_BATCH)      # at these widths AUTO keeps the one-workgroup kernel up to 16384 sequences
    assert _choice(den, B, L) == _lib.LAYER_KERNEL_BATCH
    x, _, _ = _inputs(B, Cc, L, n_num, seed=B)
    film = den.time_film(cu(np.array([321]), dev))
    out = {}
    for mode in (_lib.HEAD_FUSION_ON, _lib.HEAD_FUSION_OFF):
        fusion(mode)
        out[mode] = den.forward(cu(x, dev), film=film).cpu().numpy()
        assert np.isfinite(out[mode]).all()
    assert np.abs(out[_lib.HEAD_FUSION_ON] - out[_lib.HEAD_FUSION_OFF]).max() < 1e-3
    rows = np.unique(np.concatenate([np.arange(0, 33), np.arange(0, B, 509), np.arange(B - 40, B)]))
    y16 = orc.Denoiser(sd, bf16=True).forward(x[rows], np.full(len(rows), 321))
    assert np.abs(out[_lib.HEAD_FUSION_ON][rows] - y16).max() < 2e-2


# ---------------------------------------------------------------------------------------------
# the cluster-persistent kernel (rnn_cluster.hip): small batches at hidden 193..256
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("gru", [0, 1])
@pytest.mark.parametrize("shape", ["bi_c14", "uni_c23"])
@pytest.mark.parametrize("H,B,L", [(256, 37, 7), (250, 150, 5), (256, 32, 33)])
def test_cluster_family_forced(H, B, L, gru, shape, dev, knob):
    """Every instantiation of td_rnn_cluster_kernel (layer-0 input k-steps 1 / 2, one / two directions, LSTM / GRU,
    inference / saving): 8 workgroups per batch block and direction exchange h_t through L2 every step.  Forward against
    the bf16-emulating oracle, saving forward == inference forward bit for bit, and the cluster BPTT sweep
    (td_bptt_cluster_kernel<8, LSTM / GRU>: partial sums of W_hh^T d a exchanged through L2) on the values this kernel
    saved against the fp32 oracle's gradients."""
    from timediff_amd import _lib
    knob(_lib.LAYER_KERNEL_CLUSTER)
    dirs, Cc, n_num = (2, 14, 4) if shape == "bi_c14" else (1, 23, 7)
    sd = random_state(H, Cc, 3, dirs, gru, seed=H + 10 * gru + dirs + L)
    den = den_kernels(sd, dev)
    assert _choice(den, B, L) == _lib.LAYER_KERNEL_CLUSTER
    x, t, rng = _inputs(B, Cc, L, n_num, seed=L)
    gy = (rng.standard_normal((B, Cc, L)) / (B * L)).astype(np.float32)
    _check_forward_and_grads(den, sd, x, t, gy, np.arange(B), dev)


@pytest.mark.parametrize("gru,dirs,B", [(0, 2, 600), (1, 2, 1300), (0, 1, 1100), (1, 1, 2100)])
def test_cluster_several_blocks_per_cluster(gru, dirs, B, dev, knob):
    """More batch blocks than clusters fit the chip (16 per direction bidirectional, 32 unidirectional): the pipelined
    kernel, in which a cluster works its blocks one after the other within a step (some clusters one block fewer than
    others), state in the L2-resident scratch.  Oracle on a subset of sequences, gradients with grad_y non-zero there."""
    from timediff_amd import _lib
    knob(_lib.LAYER_KERNEL_CLUSTER)
    L, Cc, n_num = 6, 14, 4
    sd = random_state(256, Cc, 3, dirs, gru, seed=B + gru)
    den = den_kernels(sd, dev)
    assert _choice(den, B, L) == _lib.LAYER_KERNEL_CLUSTER
    x, t, rng = _inputs(B, Cc, L, n_num, seed=B)
    rows = np.unique(np.concatenate([np.arange(0, 40), np.arange(0, B, 37), np.arange(B - 48, B)]))
    gy = np.zeros((B, Cc, L), np.float32)
    gy[rows] = (rng.standard_normal((len(rows), Cc, L)) / (len(rows) * L)).astype(np.float32)
    _check_forward_and_grads(den, sd, x, t, gy, rows, dev)


@pytest.mark.parametrize("T", [1, 2, 5, 11])
def test_single_layer_cluster_chain_graph_equals_stepwise(T, dev):
    """One recurrent layer at the cluster family's shape: a captured reverse step would hold a single cluster launch, whose
    hand-off tags carry the launch epoch baked into the graph.  The chain captures two steps per graph below eight (and
    none for T = 1), so that consecutive launches on one exchange buffer never share an epoch: replayed chains equal the
    eager chain bit for bit, also when the same cached graph runs a second chain."""
    from timediff_amd import _lib
    from timediff_amd.ops import DiffusionKernels, SamplerKernels
    H, L, B = 256, 9, 70
    num, cat, K = [0, 2, 4, 6], [1, 3, 5, 7, 8], [2] * 5
    sd = random_state(H, 14, 1, 2, 0, seed=5)
    den = den_kernels(sd, dev)
    assert _choice(den, B, L) == _lib.LAYER_KERNEL_CLUSTER
    tables = {k: torch.from_numpy(v) for k, v in orc.schedule(T, "cosine").items()}
    sampler = SamplerKernels(den, DiffusionKernels(num, cat, K, L, T, tables, 9, dev))
    eager = sampler.sample(B, seed=3, patient0=10, use_graph=False)
    assert torch.isfinite(eager).all()
    for _ in range(3):
        assert torch.equal(sampler.sample(B, seed=3, patient0=10), eager)


def test_cluster_dispatch_and_agreement(dev, knob):
    """AUTO takes the cluster kernel where it can run (hidden 193..256, at most 2048 sequences bidirectional, at least 4
    steps; its one-workgroup form at hidden <= 96: test_resident_family_forced) and nothing else; its results agree with the unit-split kernel's within the bf16 bound, run to run
    bit-identically; forcing it on a shape it cannot run is an error, not a fallback."""
    from timediff_amd import _lib
    sd = random_state(256, 14, 3, 2, 0, seed=3)
    den = den_kernels(sd, dev)
    assert _choice(den, 100, 272) == _lib.LAYER_KERNEL_CLUSTER and _choice(den, 2048, 272) == _lib.LAYER_KERNEL_CLUSTER
    assert _choice(den, 2049, 272) == _lib.LAYER_KERNEL_UNITS and _choice(den, 100, 3) == _lib.LAYER_KERNEL_UNITS
    x, t, _ = _inputs(300, 14, 9, 4, seed=9)
    out = {}
    for mode in (_lib.LAYER_KERNEL_CLUSTER, _lib.LAYER_KERNEL_UNITS):
        knob(mode)
        a = den.forward(cu(x, dev), cu(t, dev)).cpu().numpy()
        b = den.forward(cu(x, dev), cu(t, dev)).cpu().numpy()
        np.testing.assert_array_equal(a, b)
        out[mode] = a
    assert np.isfinite(out[_lib.LAYER_KERNEL_CLUSTER]).all()
    assert np.abs(out[_lib.LAYER_KERNEL_CLUSTER] - out[_lib.LAYER_KERNEL_UNITS]).max() < BF16_EMUL_ATOL
    knob(_lib.LAYER_KERNEL_CLUSTER)
    assert _choice(den, 600, 9) == _lib.LAYER_KERNEL_CLUSTER and _choice(den, 600, 3) == -1
    with pytest.raises(_lib.TimediffError, match="cluster kernel was forced"):
        den.forward(cu(np.zeros((6, 14, 3), np.float32), dev), cu(np.zeros(6, np.int64), dev))
    mid = den_kernels(random_state(128, 14, 2, 2, 0, seed=1), dev)       # 4 unit blocks: neither form of the cluster kernel
    with pytest.raises(_lib.TimediffError, match="cluster kernel was forced"):
        mid.forward(cu(np.zeros((8, 14, 9), np.float32), dev), cu(np.zeros(8, np.int64), dev))


@pytest.mark.parametrize("gru", [0, 1])
def test_cluster_bptt_full_length_and_agreement(gru, dev, knob):
    """The reference's training shape (micro-batch 32, L = 272, 3 bidirectional layers, H = 256) through the cluster BPTT
    sweep: 271 hand-offs of bf16 partial sums per layer.  Gradients against the fp32 oracle (3 % of each tensor's
    largest entry, as everywhere), and against td_bptt_sweep_kernel on the same saved values (both round d a to bf16;
    the cluster also rounds its 8 partial sums of d h: 1 % of the largest entry)."""
    from timediff_amd import _lib
    B, L, Cc, n_num = 32, 272, 14, 4
    sd = random_state(256, Cc, 3, 2, gru, seed=41 + gru)
    den = den_kernels(sd, dev)
    x, t, rng = _inputs(B, Cc, L, n_num, seed=77)
    gy = (rng.standard_normal((B, Cc, L)) / (B * L)).astype(np.float32)
    film = den.time_film(cu(t, dev))
    grads = {}
    for mode in (_lib.LAYER_KERNEL_CLUSTER, _lib.LAYER_KERNEL_UNITS):
        knob(mode)
        den.forward_train(cu(x, dev), film)
        g_layer, g_misc, _ = den.backward(cu(gy, dev), film)
        grads[mode] = [g.cpu().numpy().copy() for g in g_layer]
    _, _, ref = orc.Denoiser(sd).backward(x, t, gy)
    for key, a, b in zip(den._lkeys, grads[_lib.LAYER_KERNEL_CLUSTER], grads[_lib.LAYER_KERNEL_UNITS]):
        assert np.isfinite(a).all(), key
        assert _rel(a, ref[key]) < 3e-2, (key, _rel(a, ref[key]))
        assert _rel(a, b) < 1e-2, (key, _rel(a, b))


def test_cluster_handoff_repeatable_under_load(dev, knob):
    """The multi-block cluster kernels hand h_t / partial sums of d h between workgroups through L2 inside one launch (flags +
    write-through stores, the forward's payload moved by LDS-DMA with the sc1 policy).  A stale read there would not crash:
    it would change a few values.  The kernels are otherwise deterministic (fixed summation order), so the same launch
    repeated must give the same bits every time -- with other streams streaming through the same L2s and CUs meanwhile
    (an idle chip hides stale reads), and with consumers that have just read the same addresses (L1-warm)."""
    from timediff_amd import _lib
    H, Cc, L, B = 256, 14, 48, 1536          # 48 batch blocks over 16 clusters per direction: 3 blocks per cluster
    sd = random_state(H, Cc, 3, 2, 0, seed=77)
    den = den_kernels(sd, dev)
    x, t, rng = _inputs(B, Cc, L, 4, seed=78)
    gy = (rng.standard_normal((B, Cc, L)) / (B * L)).astype(np.float32)
    xg, gyg = cu(x, dev), cu(gy, dev)
    film = den.time_film(cu(t, dev))
    knob(_lib.LAYER_KERNEL_CLUSTER)
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    stop = torch.zeros(1, device=dev)

    def load():
        with torch.cuda.stream(side):
            for _ in range(6):
                junk.mul_(1.0001).add_(stop)      # 256 MB streamed through every XCD's L2, twice

    ref_y = ref_g = None
    for rep in range(80):
        load()
        y = den.forward_train(xg, film)
        g_layer, _, g_film = den.backward(gyg, film)
        # the library GEMMs (W_hh of every layer, W_ih above the first) add up the sweep's d a in a fixed order; bias, FiLM and
        # first-layer input gradients go through float atomics and are left out
        keys = [k for k in den._lkeys if "weight_hh" in k or ("weight_ih" in k and "_l0" not in k)]
        got = {k: g for k, g in zip(den._lkeys, g_layer) if k in keys}
        torch.cuda.synchronize()
        if ref_y is None:
            ref_y, ref_g = y.clone(), {k: g.clone() for k, g in got.items()}
            assert torch.isfinite(ref_y).all() and all(torch.isfinite(g).all() for g in ref_g.values())
            continue
        assert torch.equal(y, ref_y), f"forward differs in repetition {rep}"
        for k in keys:
            assert torch.equal(got[k], ref_g[k]), f"the gradient of {k} differs in repetition {rep}"


def test_side_stream_weight_gradients_equal_inline(dev):
    """td_rnn_backward runs a layer's weight-gradient GEMMs on the library's side stream beside the sweeps below (small
    batches); td_set_side_gemm(0) keeps them on the caller's stream.  Same kernels, same fixed summation order: the weight
    gradients are bit-identical, three times in a row (a missing fork or join would race with the sweeps / the read-back)."""
    from timediff_amd import _lib
    B, L, Cc, n_num = 64, 40, 14, 4
    sd = random_state(256, Cc, 3, 2, 0, seed=12)
    den = den_kernels(sd, dev)
    x, t, rng = _inputs(B, Cc, L, n_num, seed=13)
    gy = (rng.standard_normal((B, Cc, L)) / (B * L)).astype(np.float32)
    film = den.time_film(cu(t, dev))
    keys = [k for k in den._lkeys if "weight" in k and not k.endswith("ih_l0") and not k.endswith("ih_l0_reverse")]
    got = {}
    try:
        for on in (1, 0, 1, 0, 1):
            _lib.check(_lib.lib().td_set_side_gemm(on))
            den.forward_train(cu(x, dev), film)
            g_layer, _, _ = den.backward(cu(gy, dev), film)
            cur = {k: g.clone() for k, g in zip(den._lkeys, g_layer) if k in keys}
            torch.cuda.synchronize()
            if not got:
                got = cur
                assert all(torch.isfinite(v).all() and v.abs().max() > 0 for v in got.values())
            else:
                for k in keys:
                    assert torch.equal(cur[k], got[k]), (k, on)
    finally:
        _lib.lib().td_set_side_gemm(1)
