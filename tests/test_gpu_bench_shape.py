"""The kernel bench.py times, at the length bench.py times it.

bench.py runs td_rnn_layer_kernel<8, 1, L0>, <8, 32, upper> and <8, 32, upper, HEAD> (rnn_layer.h: the batch-block family)
at H = 256, 3 bidirectional layers, C_io = 14, L = 272.  AUTO sends only batches above 8192 sequences there, where an
oracle run is minutes, so the family is forced (td_set_layer_kernel) at B = 150 -- five batch blocks, the second workgroup
with one live wave whose block is ragged (22 sequences) and three clamped ones -- and 272 recurrent steps of bf16 h go
through the kernel's register / AGPR state and the per-step LDS input staging against the bf16-emulating oracle:
hidden state and output (plain head), the HEAD-fused form against the oracle and against the unfused head, saving forward
== inference forward, parameter gradients against the fp32 oracle.  Reference semantics: blocks.py:282-297, :333-341.
Plus BASELINE.json configs[3] at real length (H = 512, L = 256) for the wide kernels.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from test_gpu_kernel_families import _check_forward_and_grads, _choice, fusion, knob  # noqa: F401 (fixtures)
from test_gpu_parity import BF16_EMUL_ATOL, _rel, cu, den_kernels
from test_gpu_wide import _inputs, random_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


# sequences the oracle is run on: the first block, a few of every other block, the whole ragged tail
ROWS = np.unique(np.concatenate([np.arange(0, 32), np.arange(32, 128, 7), np.arange(128, 150)]))


@pytest.mark.parametrize("gru", [0, 1])
def test_batch_family_at_bench_length(gru, dev, knob):
    from timediff_amd import _lib
    knob(_lib.LAYER_KERNEL_BATCH)
    H, Cc, n_num, B, L = 256, 14, 4, 150, 272
    sd = random_state(H, Cc, 3, 2, gru, seed=272 + gru)
    den = den_kernels(sd, dev)
    assert _choice(den, B, L) == _lib.LAYER_KERNEL_BATCH
    x, t, rng = _inputs(B, Cc, L, n_num, seed=272)
    gy = np.zeros((B, Cc, L), np.float32)
    gy[ROWS] = (rng.standard_normal((len(ROWS), Cc, L)) / (len(ROWS) * L)).astype(np.float32)
    _check_forward_and_grads(den, sd, x, t, gy, ROWS, dev)


@pytest.mark.parametrize("gru", [0, 1])
def test_fused_head_at_bench_length(gru, dev, knob, fusion):
    """The sampler's top-layer form (one FiLM row for the batch, the head inside the layer kernel) over 272 steps."""
    from timediff_amd import _lib
    knob(_lib.LAYER_KERNEL_BATCH)
    H, Cc, n_num, B, L = 256, 14, 4, 150, 272
    sd = random_state(H, Cc, 3, 2, gru, seed=300 + gru)
    den = den_kernels(sd, dev)
    x, _, _ = _inputs(B, Cc, L, n_num, seed=301)
    film = den.time_film(cu(np.array([321]), dev))
    out = {}
    for mode in (_lib.HEAD_FUSION_ON, _lib.HEAD_FUSION_OFF):
        fusion(mode)
        out[mode] = den.forward(cu(x, dev), film=film).cpu().numpy()
        assert np.isfinite(out[mode]).all()
    assert np.abs(out[_lib.HEAD_FUSION_ON] - out[_lib.HEAD_FUSION_OFF]).max() < 1e-3
    y16 = orc.Denoiser(sd, bf16=True).forward(x[ROWS], np.full(len(ROWS), 321))
    for mode in out:
        assert np.abs(out[mode][ROWS] - y16).max() < 2e-2, (mode, np.abs(out[mode][ROWS] - y16).max())


def test_wide_hidden_at_real_length(dev):
    """BASELINE.json configs[3] (hundreds of steps x wide hidden): H = 512, L = 256, B = 64, HiRID channel map, forward
    and gradients through rnn_wide.hip and the wide BPTT sweep."""
    H, Cc, n_num, B, L = 512, 9, 7, 64, 256
    sd = random_state(H, Cc, 3, 2, 0, seed=512)
    den = den_kernels(sd, dev)
    x, t, rng = _inputs(B, Cc, L, n_num, seed=513)
    rows = np.unique(np.concatenate([np.arange(0, 12), np.arange(30, 36), np.arange(56, 64)]))
    gy = np.zeros((B, Cc, L), np.float32)
    gy[rows] = (rng.standard_normal((len(rows), Cc, L)) / (len(rows) * L)).astype(np.float32)
    _check_forward_and_grads(den, sd, x, t, gy, rows, dev)
