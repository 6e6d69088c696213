"""The BRNN denoiser with the reference's module interface (models/ETDiff/blocks.py:12-24, :264-341).

Same class names, constructor signatures, forward signature and state_dict keys
(`rnn.rnn.weight_ih_l0[_reverse]`, `rnn.time_mlp.1.*`, `rnn.norm.*`, `time_emb.{1,3}.*`, `fc.*`), so a
reference checkpoint loads unchanged.  The torch sub-modules below are parameter containers only: the
arithmetic runs in the HIP kernels behind `timediff_amd.ops.DenoiserKernels`.
"""
import math

import torch
import torch.nn as nn

from .ops import DenoiserKernels


class SinusoidalPosEmb(nn.Module):
    """blocks.py:12-24 (kept as a module so `time_emb.1` / `time_emb.3` keep their indices)."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        half = self.dim // 2
        freq = torch.exp(torch.arange(half, device=x.device) * -(math.log(10000) / (half - 1)))
        ang = x[:, None] * freq[None, :]
        return torch.cat((ang.sin(), ang.cos()), dim=-1)


class RNNBlock(nn.Module):
    """blocks.py:264-297: holds nn.LSTM / nn.GRU, the FiLM linear and the LayerNorm parameters."""

    def __init__(self, input_channels, output_channels, layers, batch_first, dropout, bidirectional, time_dim, model):
        super().__init__()
        if model == "lstm":
            self.rnn = nn.LSTM(input_channels, output_channels, layers, batch_first=batch_first, dropout=dropout,
                               bidirectional=bidirectional)
        elif model == "gru":
            self.rnn = nn.GRU(input_channels, output_channels, layers, batch_first=batch_first, dropout=dropout,
                              bidirectional=bidirectional)
        else:
            raise ValueError(f"Invalid model name \"{model}\"! It can be only \"lstm\" or \"gru\"!")
        width = output_channels * 2 if bidirectional else output_channels
        self.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_dim, width * 2))
        self.norm = nn.LayerNorm(width)


class RNN(nn.Module):
    def __init__(self, input_channels, hidden_channels, output_channels, layers, model="lstm", dropout=0,
                 bidirectional=True, self_condition=False, embed_dim=64, time_dim=256):
        super().__init__()
        if not 0 <= dropout < 1:
            raise ValueError(f"dropout should be a number in range [0, 1), got {dropout}")
        self.dropout = float(dropout) if layers > 1 else 0.0      # nn.LSTM: no effect (a warning) with one layer
        if self_condition:
            raise NotImplementedError("self_condition is never enabled by the reference's CLI (etdiff_train.py:224, :245)")
        self.self_condition = self_condition
        self.rnn = RNNBlock(input_channels, hidden_channels, layers, time_dim=time_dim, model=model, batch_first=True,
                            dropout=dropout, bidirectional=bidirectional)
        self.time_emb = nn.Sequential(SinusoidalPosEmb(embed_dim), nn.Linear(embed_dim, time_dim), nn.GELU(),
                                      nn.Linear(time_dim, time_dim))
        self.fc = nn.Linear(hidden_channels * 2 if bidirectional else hidden_channels, output_channels)
        self._kernels = None
        self._kernels_version = None

    # -- kernel plumbing -------------------------------------------------------------------------
    def _param_version(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def kernels(self) -> DenoiserKernels:
        """The HIP-side view of the parameters; the packed MFMA weight cache is rebuilt whenever a parameter
        was modified in place (optimizer step, load_state_dict) or moved."""
        version = self._param_version()
        if self._kernels is None or self._kernels_version != version:
            sd = {k: v for k, v in self.named_parameters()}
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("the denoiser runs on the MI355X only: move the module to a ROCm device (.cuda())")
            if self._kernels is None or self._kernels.device != dev:
                self._kernels = DenoiserKernels(sd)
            else:
                self._kernels._sd = sd
                self._kernels.refresh()
            self._kernels_version = version
        return self._kernels

    def __deepcopy__(self, memo):
        # the EMA wrapper deep-copies the diffusion module: ctypes handles do not copy, the cache is rebuilt
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        import copy
        for k, v in self.__dict__.items():
            setattr(new, k, None if k in ("_kernels", "_kernels_version", "_grad_sink") else copy.deepcopy(v, memo))
        return new

    def forward(self, x, time, x_self_cond=None):
        """x: (B, C, L) fp32, time: (B,) int64 -> (B, C_out, L)."""
        from .autograd import denoiser_apply
        return denoiser_apply(self, x, time)


# Names the reference's CLI imports from this module (etdiff_train.py:18) but never instantiates.
class NeuralCDE(nn.Module):
    def __init__(self, *a, **kw):
        raise NotImplementedError("NeuralCDE is dead code in the reference (never instantiated); not part of the hot path")


class EncoderDecoderRNN(nn.Module):
    def __init__(self, *a, **kw):
        raise NotImplementedError("EncoderDecoderRNN is dead code in the reference (never instantiated)")
