"""The TSTR (train-on-synthetic, test-on-real) RNN classifier with the reference's interface
(evaluate/nn_classifier.py:89-171: `RNN`, `nnTimeSeriesClassifier`, `normalize_data`, `standardize_data`), on the
recurrent kernels of the denoiser: forward with saves, BPTT and weight gradients are td_cls_forward / td_cls_backward
(csrc/rnn_train.hip); the update is AdamW on one flat fp32 buffer (decoupled decay + td_clip_adam_ema).

Same constructor signatures, `forward(X)` with X (B, C, L) -> P(y = 1) of shape (B,), `train()`, `predict_proba(X)`,
and the same state_dict keys (`lstm.*` / `gru.*`, `final.0.*`), so weights move between the two implementations.  The
torch sub-modules are parameter containers only.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, lib, ptr


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def normalize_data(data: torch.Tensor, eps=0):
    """min-max per channel over (batch, time) (nn_classifier.py:27-30) -> (data, min, max)"""
    lo, hi = data.amin(dim=(0, 2))[None, :, None], data.amax(dim=(0, 2))[None, :, None]
    return ((data - lo + eps) / (hi - lo + 2 * eps)).float(), lo.float(), hi.float()


def standardize_data(data: torch.Tensor):
    """zero mean / unit (unbiased) standard deviation per channel (nn_classifier.py:32-35) -> (data, mean, std)"""
    mean, std = data.mean(dim=(0, 2))[None, :, None], data.std(dim=(0, 2))[None, :, None]
    return ((data - mean) / std).float(), mean.float(), std.float()


class _ClsFn(torch.autograd.Function):
    """prob = RNN(x): one autograd node; backward takes d loss / d prob."""

    @staticmethod
    def forward(ctx, module, x, *params):
        prob = module._forward_kernels(x)
        ctx.module = module
        ctx.save_for_backward(prob)
        return prob

    @staticmethod
    def backward(ctx, grad_prob):
        (prob,) = ctx.saved_tensors
        grads = ctx.module._backward_kernels(prob, grad_prob=grad_prob.contiguous())
        return (None, None, *grads)


class RNN(nn.Module):
    """nn_classifier.py:89-127."""

    def __init__(self, in_channels, hidden_dim=64, model="lstm"):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.model = model
        if model == "lstm":
            self.lstm = nn.LSTM(in_channels, hidden_dim, bidirectional=True, batch_first=True)
        elif model == "gru":
            self.gru = nn.GRU(in_channels, hidden_dim, bidirectional=True, batch_first=True)
        else:
            raise ValueError("Not available")
        self.final = nn.Sequential(nn.Linear(hidden_dim * 2, 1), nn.Sigmoid())
        self.cfg = _lib.RnnConfig(c_in=in_channels, hidden=hidden_dim, c_out=1, layers=1, dirs=2, gru=int(model == "gru"),
                                  embed_dim=64, time_dim=256)
        self._packed = self._packT = self._ws = None
        self._shape = None

    # -- kernel plumbing ----------------------------------------------------------------------------
    def _rnn(self):
        return self.lstm if self.model == "lstm" else self.gru

    def _layer_params(self):
        r = self._rnn()
        return [getattr(r, f"{n}_l0{sfx}") for sfx in ("", "_reverse") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]

    def _table(self, tensors):
        return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def _forward_kernels(self, x):
        dev = self.final[0].weight.device
        if dev.type != "cuda":
            raise RuntimeError("the classifier runs on the MI355X only: move the module to a ROCm device (.cuda())")
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        B, Cin, L = x.shape
        assert Cin == self.cfg.c_in, f"expected {self.cfg.c_in} channels, got {Cin}"
        lw = self._layer_params()
        for t in lw:
            assert t.is_contiguous() and t.dtype == torch.float32
        if self._packed is None or self._packed.device != dev:
            self._packed = torch.empty(lib().td_rnn_packed_bytes(C.byref(self.cfg)), dtype=torch.uint8, device=dev)
            self._packT = torch.empty(lib().td_rnn_packT_bytes(C.byref(self.cfg)), dtype=torch.uint8, device=dev)
        # the weights may have moved since the last call (optimizer step): the two pack kernels take microseconds
        check(lib().td_rnn_pack_weights(C.byref(self.cfg), self._table(lw), ptr(self._packed), _stream()))
        need = lib().td_cls_workspace_bytes(C.byref(self.cfg), B, L)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        prob = torch.empty(B, dtype=torch.float32, device=dev)
        lin = self.final[0]
        check(lib().td_cls_forward(C.byref(self.cfg), ptr(self._packed), ptr(x), ptr(lin.weight), ptr(lin.bias), B, L, ptr(prob),
                                   ptr(self._ws), _stream()))
        self._shape = (B, L)
        return prob

    def _backward_kernels(self, prob, y=None, grad_prob=None, sink=None):
        """Gradients of the last forward: of mean BCE(prob, y) when y is given, else through d loss / d prob.
        -> [8 layer tensors, final weight, final bias]; sink: the same list of tensors to ACCUMULATE into instead."""
        B, L = self._shape
        lw = self._layer_params()
        lin = self.final[0]
        grads = sink if sink is not None else [torch.zeros_like(t) for t in lw + [lin.weight, lin.bias]]
        check(lib().td_rnn_pack_weights_T(C.byref(self.cfg), self._table(lw), ptr(self._packT), _stream()))
        check(lib().td_cls_backward(C.byref(self.cfg), ptr(self._packT), ptr(lin.weight), ptr(prob), ptr(y), ptr(grad_prob), B, L,
                                    ptr(self._ws), self._table(grads[:8]), ptr(grads[8]), ptr(grads[9]), _stream()))
        return grads

    def forward(self, X):
        params = self._layer_params() + [self.final[0].weight, self.final[0].bias]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _ClsFn.apply(self, X, *params)
        return self._forward_kernels(X)


class nnTimeSeriesClassifier:
    """nn_classifier.py:129-171: standardise, AdamW (torch defaults: lr 1e-3, betas (0.9, 0.999), eps 1e-8, weight decay
    1e-2), binary cross entropy, `train_num_steps` steps over shuffled batches of `bsz`."""

    def __init__(self, model, X_train, y_train, bsz, check_point_interval=100, loss_func="cross entropy", normalize="standardize",
                 wandb=None, train_num_steps=5000):
        assert isinstance(X_train, torch.Tensor), "X_train must be a torch.Tensor!"
        assert isinstance(y_train, torch.Tensor), "y_train must be a torch.Tensor!"
        assert len(X_train.shape) == 3, "X_train must be 3D tensor!"
        assert len(y_train.shape) == 1, "y_train must be 1D tensor!"
        assert X_train.shape[0] == len(y_train), "X and y must have the same number of samples!"
        assert torch.equal(torch.unique(y_train), torch.tensor([0, 1])), "y must be binary and only (0,1)!"
        if loss_func != "cross entropy":
            raise ValueError("only the reference's \"cross entropy\" loss exists")
        if not torch.cuda.is_available():
            raise RuntimeError("the classifier trains on the MI355X only (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.model = model(in_channels=X_train.shape[1]).to(self.device)
        if normalize == "standardize":
            X_train, _, _ = standardize_data(X_train)
        elif normalize == "normalize":
            X_train, _, _ = normalize_data(X_train)
        self.X_train, self.y_train = X_train.float().to(self.device), y_train.float().to(self.device)
        self.bsz = bsz
        self.wandb = wandb
        self.check_point_interval = check_point_interval
        self.train_num_steps = train_num_steps
        # flat parameter / gradient / moment buffers (the order of _backward_kernels' list)
        self.params = self.model._layer_params() + [self.model.final[0].weight, self.model.final[0].bias]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=self.device)
        self.flat_grad = torch.zeros_like(self.flat)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.grads, off = [], 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            self.grads.append(self.flat_grad[off:off + k].view_as(p))
            off += k
        self.lr, self.betas, self.eps, self.weight_decay = 1e-3, (0.9, 0.999), 1e-8, 1e-2
        self.step = 0
        self._perm, self._pos = None, 0
        self.last_loss = None

    def _next_batch(self):
        """DataLoader(shuffle=True) over the resident tensors, cycled (nn_classifier.py:14-17, :153)."""
        n = self.X_train.shape[0]
        if self._perm is None or self._pos >= n:
            self._perm, self._pos = torch.randperm(n, device=self.device), 0
        idx = self._perm[self._pos:self._pos + self.bsz]
        self._pos += self.bsz
        return self.X_train[idx], self.y_train[idx]

    def train_step(self):
        X, y = self._next_batch()
        prob = self.model._forward_kernels(X)
        self.flat_grad.zero_()
        self.model._backward_kernels(prob, y=y.contiguous(), sink=self.grads)
        self.step += 1
        self.flat.mul_(1.0 - self.lr * self.weight_decay)            # AdamW's decoupled decay
        check(lib().td_clip_adam_ema(ptr(self.flat), ptr(self.flat_grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), None,
                                     self.flat.numel(), None, 0.0, self.lr, self.betas[0], self.betas[1], self.eps, self.step, 0,
                                     0.0, _stream()))
        self.last_prob, self.last_y = prob, y

    def train(self):
        while self.step < self.train_num_steps:
            self.train_step()

    def change_device(self, device):
        """nn_classifier.py:163-165 moves the model.  Here the model's parameters are views of flat buffers on the device the
        classifier was built on, the kernels have no CPU form, and nothing in the reference calls this with another
        accelerator: the current device is accepted, anything else is refused rather than half-done."""
        target = torch.device(device)
        if target.type != "cuda":
            raise RuntimeError("the classifier runs on the MI355X only")
        if target.index is not None and target.index != (self.device.index if self.device.index is not None else torch.cuda.current_device()):
            raise NotImplementedError(f"the classifier was built on {self.device}; build a new one on {target} (its parameters are "
                                      "views of flat device buffers)")

    @torch.no_grad()
    def predict_proba(self, X):
        out = []
        for i in range(0, X.shape[0], 4096):
            out.append(self.model(X[i:i + 4096].to(self.device)).cpu())
        return torch.cat(out).numpy()
