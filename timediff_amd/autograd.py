"""Autograd wiring of the denoiser: ONE autograd node whose forward and backward run entirely in the HIP library --
the time-embedding MLP + FiLM linear (td_rnn_time_film / td_rnn_time_film_backward), the recurrent stack, the
LayerNorm/FiLM head and fc (td_rnn_forward_train / td_rnn_backward).  No stock torch op is on the training path."""
import torch

from . import _lib


class _DenoiserFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, time, *params):
        k = module.kernels()
        film = k.time_film(time)
        # inter-layer dropout (training mode only, as nn.LSTM): a fresh 64-bit seed per call from torch's generator, so
        # torch.manual_seed makes a run repeatable; the backward pass re-sets the same (p, seed)
        p = module.dropout if module.training else 0.0
        ctx.dropout = (p, int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0)
        # (p, seed) is process-wide state of the library that also sizes the training workspace: it holds for this call
        # only, so that a later direct forward_train / backward caller (tests, the bench, C-ABI users) never inherits it
        k.set_dropout(*ctx.dropout)
        try:
            y = k.forward_train(x, film)
        finally:
            k.set_dropout(0.0)
        ctx.module = module
        ctx.save_for_backward(film, time)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        film, time = ctx.saved_tensors
        k = ctx.module.kernels()
        k.set_dropout(*ctx.dropout)
        try:
            return _DenoiserFn._backward_body(ctx, k, grad_y, film, time)
        finally:
            k.set_dropout(0.0)

    @staticmethod
    def _backward_body(ctx, k, grad_y, film, time):
        sink = getattr(ctx.module, "_grad_sink", None)
        if sink is not None:
            # the trainer's flat gradient buffer: the kernels accumulate straight into the .grad views (no
            # AccumulateGrad pass), and the trainer may start a layer's all-reduce while the layers below are
            # still being differentiated (et_diff.py: ETDiff._layer_reduce)
            # (one call for the recurrent stack, the head AND the time-embedding MLP: the latter runs beside the BPTT sweeps)
            k.backward(grad_y.contiguous(), film, sink=sink["tensors"], on_layer_done=sink.get("on_layer_done"), time=time)
            return (None, None, None) + (None,) * (len(k._lkeys) + len(_lib.MISC_KEYS))
        g_layer, g_misc, _ = k.backward(grad_y.contiguous(), film, time=time)
        return (None, None, None, *g_layer, *[g_misc[key] for key in _lib.MISC_KEYS])


def _train_params(module):
    k = module.kernels()
    named = dict(module.named_parameters())
    return [named[key] for key in k._lkeys] + [named[key] for key in _lib.MISC_KEYS]


def denoiser_apply(module, x, time):
    """RNN.forward (blocks.py:333-341)."""
    params = _train_params(module) if torch.is_grad_enabled() else None
    if params is not None and any(p.requires_grad for p in module.parameters()):
        return _DenoiserFn.apply(module, x, time, *params)
    return module.kernels().forward(x, time)
