"""Autograd wiring of the denoiser: the recurrent stack, LayerNorm/FiLM head and fc run forward and
backward in the HIP library; only the time-embedding MLP (three (B, 256)-sized linears, blocks.py:324-329
and :285) is left to torch's own autograd during training."""
import torch

from . import _lib


class _DenoiserFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, film, *params):
        k = module.kernels()
        y = k.forward_train(x, film)
        ctx.module = module
        ctx.save_for_backward(film)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        (film,) = ctx.saved_tensors
        k = ctx.module.kernels()
        sink = getattr(ctx.module, "_grad_sink", None)
        if sink is not None:
            # the trainer's flat gradient buffer: the kernels accumulate straight into the .grad views (no
            # AccumulateGrad pass), and the trainer may start a layer's all-reduce while the layers below are
            # still being differentiated (et_diff.py: ETDiff._layer_reduce)
            _, _, g_film = k.backward(grad_y.contiguous(), film, sink=sink["tensors"], on_layer_done=sink.get("on_layer_done"))
            return (None, None, g_film) + (None,) * (len(k._lkeys) + len(_lib.MISC_KEYS[6:]))
        g_layer, g_misc, g_film = k.backward(grad_y.contiguous(), film)
        tail = [g_misc[key] for key in _lib.MISC_KEYS[6:]]
        return (None, None, g_film, *g_layer, *tail)


def _train_params(module):
    k = module.kernels()
    named = dict(module.named_parameters())
    return [named[key] for key in k._lkeys] + [named[key] for key in _lib.MISC_KEYS[6:]]


def denoiser_apply(module, x, time):
    """RNN.forward (blocks.py:333-341)."""
    params = _train_params(module) if torch.is_grad_enabled() else None
    if params is not None and any(p.requires_grad for p in module.parameters()):
        film = module.rnn.time_mlp(module.time_emb(time))          # torch autograd: tiny (B, 256) linears
        return _DenoiserFn.apply(module, x, film, *params)
    return module.kernels().forward(x, time)
