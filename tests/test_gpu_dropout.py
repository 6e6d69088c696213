"""Inter-layer dropout of the recurrent stack (`dropout=` of blocks.py:268-270, etdiff_train.py --dropout) on the HIP
training path, against the CPU oracle fed the SAME masks (td_rnn_dropout_mask exports the mask the kernels' counter-based
generator produces; tests/test_oracle_dropout.py pins the oracle's masked forward / backward against torch autograd).
Tolerances as in test_gpu_parity.py: 2e-2 on y against the bf16-emulating oracle, 3 % of each gradient's largest entry."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from test_gpu_parity import _rel, cu, den_kernels
from test_gpu_wide import _inputs, random_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def no_dropout_left_behind():
    from timediff_amd.ops import DenoiserKernels
    yield
    DenoiserKernels.set_dropout(0.0, 0)


# (H, layers, dirs, gru, C, n_num, B, L, p)
CASES = [
    (92, 3, 2, 0, 14, 4, 37, 20, 0.25),     # the reference's LSTM shape, ragged batch
    (64, 2, 1, 1, 9, 7, 33, 11, 0.5),       # GRU, unidirectional
    (256, 3, 2, 0, 14, 4, 70, 12, 0.1),     # full-width kernels (8 unit blocks)
    (512, 2, 2, 0, 14, 4, 37, 8, 0.3),      # wide-hidden kernels
    (48, 4, 2, 1, 9, 7, 300, 9, 0.2),       # GRU, four layers, ten batch blocks
]


@pytest.mark.parametrize("H,layers,dirs,gru,C,n_num,B,L,p", CASES)
def test_dropout_forward_backward_against_oracle(H, layers, dirs, gru, C, n_num, B, L, p, dev):
    sd = random_state(H, C, layers, dirs, gru, seed=H + 3)
    den = den_kernels(sd, dev)
    x, t, rng = _inputs(B, C, L, n_num, seed=H + 2)
    gy = (rng.standard_normal((B, C, L)) / (B * L)).astype(np.float32)
    film = den.time_film(cu(t, dev))
    y_plain = den.forward_train(cu(x, dev), film).cpu().numpy()
    den.set_dropout(p, seed=0x1234567 + H)
    masks = np.stack([den.dropout_mask(B, L, l).cpu().numpy() for l in range(layers - 1)])
    pq = round(p * 65536) / 65536
    assert set(np.unique(masks).tolist()) <= {0.0, float(np.float32(1.0 / (1.0 - pq)))}
    drop = float((masks == 0).mean())
    assert abs(drop - p) < 4 * np.sqrt(p * (1 - p) / masks.size) + 1e-3, (drop, p)
    if layers > 2:
        assert (masks[0] != masks[1]).mean() > 0.05          # every layer draws its own
    y = den.forward_train(cu(x, dev), film).cpu().numpy()
    assert np.abs(y - y_plain).max() > 1e-2
    y16 = orc.Denoiser(sd, bf16=True).forward(x, t, masks=masks)
    assert np.abs(y - y16).max() < 2e-2, np.abs(y - y16).max()
    g_layer, g_misc, g_film = den.backward(cu(gy, dev), film)
    _, _, ref = orc.Denoiser(sd).backward(x, t, gy, masks=masks)
    for key, got in list(zip(den._lkeys, g_layer)) + list(g_misc.items()):
        assert _rel(got.cpu().numpy(), ref[key]) < 3e-2, (key, _rel(got.cpu().numpy(), ref[key]))
    # without the masks the oracle's gradients are a different thing altogether (the check above is not vacuous)
    _, _, ref0 = orc.Denoiser(sd).backward(x, t, gy)
    k0 = den._lkeys[0]
    assert _rel(g_layer[0].cpu().numpy(), ref0[k0]) > 6e-2
    # another seed: another mask; p = 0: the plain network, bit for bit
    den.set_dropout(p, seed=99)
    assert (den.dropout_mask(B, L, 0).cpu().numpy() != masks[0]).mean() > 0.05
    den.set_dropout(0.0)
    np.testing.assert_array_equal(den.forward_train(cu(x, dev), film).cpu().numpy(), y_plain)


def test_dropout_module_modes(dev):
    """RNN(dropout=p) (blocks.py:300-331): drops in training mode only, a fresh mask per call drawn from torch's seeded
    generator; eval mode and the no-grad (sampling) path are the plain network."""
    from timediff_amd.blocks import RNN
    torch.manual_seed(0)
    net = RNN(9, 64, 9, 3, model="lstm", dropout=0.3).to(dev)
    ref = RNN(9, 64, 9, 3, model="lstm", dropout=0).to(dev)
    ref.load_state_dict(net.state_dict())
    x = torch.randn(20, 9, 15, device=dev)
    t = torch.randint(0, 1000, (20,), device=dev)
    net.train()
    torch.manual_seed(5)
    a = net(x, t)
    b = net(x, t)
    assert (a - b).abs().max() > 1e-2
    torch.manual_seed(5)
    a2 = net(x, t)
    torch.testing.assert_close(a2, a, rtol=0, atol=0)
    (a2 * torch.randn_like(a2)).sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    net.eval()
    ref.train()
    torch.testing.assert_close(net(x, t), ref(x, t), rtol=0, atol=0)
    with torch.no_grad():
        torch.testing.assert_close(net(x, t), ref(x, t), rtol=0, atol=0)
    with pytest.raises(ValueError):
        RNN(9, 64, 9, 3, dropout=1.0)
    assert RNN(9, 64, 9, 1, dropout=0.5).dropout == 0.0      # one layer: nothing to drop between
