"""The oracle's inter-layer dropout (oracle/etdiff_oracle.c: or_set_dropout_masks) pinned against torch on the CPU.

nn.LSTM / nn.GRU draw their dropout masks inside the fused op, so they cannot be handed a mask; by their documented
semantics (`dropout=`: "a Dropout layer on the outputs of each LSTM layer except the last layer") an N-layer module equals
N one-layer modules with the mask multiplied in between, which is what this builds with torch's own nn.LSTM / nn.GRU and
autograd in fp32.  The reference's RNN (blocks.py:264-341) adds LayerNorm * FiLM + fc on top; the oracle's head is
pinned by tests/test_oracle_golden.py, here it rides along through torch ops as well."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import oracle as orc


def _state(H, C, layers, dirs, gru, seed):
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
    sd["time_emb.1.weight"] = torch.randn(32, 16, generator=g) / 4; sd["time_emb.1.bias"] = 0.1 * torch.randn(32, generator=g)
    sd["time_emb.3.weight"] = torch.randn(32, 32, generator=g) / 6; sd["time_emb.3.bias"] = 0.1 * torch.randn(32, generator=g)
    sd["rnn.time_mlp.1.weight"] = torch.randn(2 * DH, 32, generator=g) / 6; sd["rnn.time_mlp.1.bias"] = torch.zeros(2 * DH)
    sd["rnn.norm.weight"] = 1 + 0.1 * torch.randn(DH, generator=g); sd["rnn.norm.bias"] = 0.1 * torch.randn(DH, generator=g)
    sd["fc.weight"] = torch.randn(C, DH, generator=g) / np.sqrt(DH); sd["fc.bias"] = 0.1 * torch.randn(C, generator=g)
    return sd


def _torch_stack(sd, x, t, masks, H, layers, dirs, gru):
    """-> (y, {name: grad}) of sum(y * gy) through one-layer torch modules with the masks in between."""
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    cls = nn.GRU if gru else nn.LSTM
    h = x.transpose(1, 2)
    for l in range(layers):
        m = cls(h.shape[-1], H, 1, batch_first=True, bidirectional=dirs == 2)
        names = [n for n, _ in m.named_parameters()]
        flat = {n: P[f"rnn.rnn.{n.replace('_l0', f'_l{l}')}"] for n in names}
        h = torch.func.functional_call(m, flat, (h,))[0]
        if l + 1 < layers:
            h = h * masks[l]
    half = 8
    freq = torch.exp(torch.arange(half) * -(np.log(10000) / (half - 1)))
    ang = t[:, None].float() * freq[None]
    e = torch.cat((ang.sin(), ang.cos()), -1)
    e = nn.functional.gelu(e @ P["time_emb.1.weight"].T + P["time_emb.1.bias"]) @ P["time_emb.3.weight"].T + P["time_emb.3.bias"]
    film = nn.functional.silu(e) @ P["rnn.time_mlp.1.weight"].T + P["rnn.time_mlp.1.bias"]
    scale, shift = film[:, None, :].chunk(2, dim=-1)
    hn = nn.functional.layer_norm(h, (h.shape[-1],), P["rnn.norm.weight"], P["rnn.norm.bias"])
    y = (hn * (scale + 1) + shift) @ P["fc.weight"].T + P["fc.bias"]
    return y.transpose(1, 2), P


@pytest.mark.parametrize("H,layers,dirs,gru", [(12, 3, 2, 0), (9, 2, 1, 1), (10, 2, 2, 1), (8, 3, 1, 0)])
def test_oracle_dropout_against_torch_layer_stack(H, layers, dirs, gru):
    torch.manual_seed(H)
    B, C, L, p = 5, 6, 7, 0.4
    sd = _state(H, C, layers, dirs, gru, seed=H + layers)
    x = torch.randn(B, C, L)
    t = torch.randint(0, 1000, (B,))
    gy = torch.randn(B, C, L)
    masks = (torch.rand(layers - 1, B, L, dirs * H) >= p).float() / (1 - p)
    y, P = _torch_stack(sd, x, t, masks, H, layers, dirs, gru)
    (y * gy).sum().backward()
    den = orc.Denoiser({k: v.numpy() for k, v in sd.items()})
    assert den.cfg.embed_dim == 16 and den.cfg.time_dim == 32
    y_or = den.forward(x.numpy(), t.numpy(), masks=masks.numpy())
    np.testing.assert_allclose(y_or, y.detach().numpy(), rtol=1e-4, atol=2e-5)
    y_b, _, grads = den.backward(x.numpy(), t.numpy(), gy.numpy(), masks=masks.numpy())
    np.testing.assert_allclose(y_b, y.detach().numpy(), rtol=1e-4, atol=2e-5)
    for k, ref in P.items():
        ref = ref.grad.numpy()
        assert np.abs(grads[k] - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), (k, np.abs(grads[k] - ref).max())
    # the masks matter, and clearing them restores the plain network
    y_plain = den.forward(x.numpy(), t.numpy())
    assert np.abs(y_plain - y_or).max() > 1e-2
    ones = np.ones_like(masks.numpy())
    np.testing.assert_array_equal(den.forward(x.numpy(), t.numpy(), masks=ones), y_plain)
