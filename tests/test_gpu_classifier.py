"""GPU parity of the TSTR classifier (evaluate/nn_classifier.py:89-171) on the recurrent kernels.

The reference class is a stock bidirectional nn.LSTM / nn.GRU + Linear + Sigmoid, so the checker here is that very
composition in fp32 torch on the CPU with the same weights (a floating-point kernel: torch fp32 is the reference;
tolerances are the denoiser's: bf16 MFMA operands, fp32 accumulation and state)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


class _TorchRNN(nn.Module):
    """evaluate/nn_classifier.py:89-127, restated for the CPU check."""

    def __init__(self, in_channels, hidden_dim, model):
        super().__init__()
        cls = nn.LSTM if model == "lstm" else nn.GRU
        setattr(self, model, cls(in_channels, hidden_dim, bidirectional=True, batch_first=True))
        self.model = model
        self.final = nn.Sequential(nn.Linear(hidden_dim * 2, 1), nn.Sigmoid())

    def forward(self, X):
        out, ht = getattr(self, self.model)(X.permute(0, 2, 1))
        ht = ht[0] if self.model == "lstm" else ht
        return self.final(ht.permute(1, 0, 2).reshape(X.shape[0], -1)).squeeze(-1)


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("model,C,H,B,L", [("lstm", 9, 64, 50, 24), ("gru", 5, 64, 33, 17), ("lstm", 23, 40, 70, 30), ("gru", 9, 256, 40, 12)])
def test_classifier_forward_backward(model, C, H, B, L, dev):
    from evaluate.nn_classifier import RNN
    torch.manual_seed(C + H)
    ref = _TorchRNN(C, H, model)
    mine = RNN(in_channels=C, hidden_dim=H, model=model)
    missing = mine.load_state_dict(ref.state_dict())          # the same keys as the reference's class
    assert not missing.missing_keys and not missing.unexpected_keys
    mine = mine.to(dev)
    X = torch.randn(B, C, L)
    y = (torch.rand(B) > 0.5).float()
    p_ref = ref(X)
    loss_ref = nn.functional.binary_cross_entropy(p_ref, y)
    loss_ref.backward()
    p = mine(X.to(dev))
    assert p.shape == (B,)
    assert np.abs(p.detach().cpu().numpy() - p_ref.detach().numpy()).max() < 1e-2
    # through autograd (d loss / d prob from torch's BCE) ...
    loss = nn.functional.binary_cross_entropy(p, y.to(dev))
    loss.backward()
    want = dict(ref.named_parameters())
    for name, q in mine.named_parameters():
        assert _rel(q.grad.cpu().numpy(), want[name].grad.numpy()) < 3e-2, (name, _rel(q.grad.cpu().numpy(), want[name].grad.numpy()))
    # ... and with the BCE derivative fused into the head kernel (what nnTimeSeriesClassifier.train uses)
    with torch.no_grad():
        prob = mine._forward_kernels(X.to(dev))
    fused = mine._backward_kernels(prob, y=y.to(dev))
    names = [n for n, _ in mine.named_parameters()]
    params = mine._layer_params() + [mine.final[0].weight, mine.final[0].bias]
    by_ptr = {q.data_ptr(): n for n, q in mine.named_parameters()}
    for g, q in zip(fused, params):
        n = by_ptr[q.data_ptr()]
        assert _rel(g.cpu().numpy(), want[n].grad.numpy()) < 3e-2, n
    assert len(names) == len(params)


def test_classifier_trains(dev):
    """nnTimeSeriesClassifier on a separable synthetic problem (the label decides the mean of one channel late in the
    sequence): the fused training step lowers the loss and the probabilities separate the classes."""
    from evaluate.nn_classifier import RNN, nnTimeSeriesClassifier
    from functools import partial
    g = torch.Generator().manual_seed(0)
    N, C, L = 512, 6, 20
    y = (torch.rand(N, generator=g) > 0.5).long()
    X = torch.randn(N, C, L, generator=g)
    X[:, 2, L // 2:] += (y.float() * 2 - 1)[:, None] * 1.5
    torch.manual_seed(1)
    clf = nnTimeSeriesClassifier(partial(RNN, hidden_dim=64, model="gru"), X, y, bsz=64, train_num_steps=150)
    p0 = clf.predict_proba(clf.X_train)
    clf.train()
    p1 = clf.predict_proba(clf.X_train)
    yb = y.numpy().astype(bool)

    def bce(p):
        p = np.clip(p, 1e-6, 1 - 1e-6)
        return float(-(np.log(p[yb]).sum() + np.log(1 - p[~yb]).sum()) / N)
    assert bce(p1) < 0.5 * bce(p0), (bce(p0), bce(p1))
    assert ((p1 > 0.5) == yb).mean() > 0.95
    assert clf.step == 150 and np.isfinite(p1).all()
