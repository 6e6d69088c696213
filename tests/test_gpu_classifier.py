"""GPU parity of the TSTR classifier (evaluate/nn_classifier.py:89-171) on the recurrent kernels.

The checker is the reference itself: fixtures recorded from its RNN class on the CPU (tests/golden/classifier_*.npz, written
by tests/golden/make_golden.py: gen_classifier).  Tolerances are the denoiser's: bf16 MFMA operands, fp32 accumulation and
state."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("name", ["classifier_lstm_c9_h64", "classifier_gru_c5_h64", "classifier_lstm_c23_h40", "classifier_gru_c9_h256"])
def test_classifier_forward_backward(name, dev):
    """Against the reference's own class: tests/golden/make_golden.py (gen_classifier) imports evaluate/nn_classifier.py
    from the reference, runs RNN.forward, F.binary_cross_entropy and backward on the CPU and records the weights, inputs,
    probabilities, loss and every parameter gradient."""
    from evaluate.nn_classifier import RNN
    with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
        g = {k: f[k] for k in f.files}
    C, H, B, L = [int(v) for v in g["cfg"]]
    model = "lstm" if "lstm" in name else "gru"
    mine = RNN(in_channels=C, hidden_dim=H, model=model)
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    missing = mine.load_state_dict(sd)          # the same keys as the reference's class
    assert not missing.missing_keys and not missing.unexpected_keys
    mine = mine.to(dev)
    X, y = torch.from_numpy(g["X"]), torch.from_numpy(g["y"])
    want = {k[5:]: v for k, v in g.items() if k.startswith("grad.")}
    p = mine(X.to(dev))
    assert p.shape == (B,)
    assert np.abs(p.detach().cpu().numpy() - g["prob"]).max() < 1e-2
    # through autograd (d loss / d prob from torch's BCE) ...
    loss = nn.functional.binary_cross_entropy(p, y.to(dev))
    assert abs(float(loss) - float(g["loss"])) < 1e-2
    loss.backward()
    assert set(want) == {n for n, _ in mine.named_parameters()}
    for n, q in mine.named_parameters():
        assert _rel(q.grad.cpu().numpy(), want[n]) < 3e-2, (n, _rel(q.grad.cpu().numpy(), want[n]))
    # ... and with the BCE derivative fused into the head kernel (what nnTimeSeriesClassifier.train uses)
    with torch.no_grad():
        prob = mine._forward_kernels(X.to(dev))
    fused = mine._backward_kernels(prob, y=y.to(dev))
    params = mine._layer_params() + [mine.final[0].weight, mine.final[0].bias]
    by_ptr = {q.data_ptr(): n for n, q in mine.named_parameters()}
    assert len(by_ptr) == len(params)
    for gr, q in zip(fused, params):
        n = by_ptr[q.data_ptr()]
        assert _rel(gr.cpu().numpy(), want[n]) < 3e-2, n


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
