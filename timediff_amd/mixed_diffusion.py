"""MixedDiffusion with the reference's interface (models/ETDiff/mixed_diffusion.py:68-454).

Constructor signature, attribute names and the 21 registered buffers (names, order, dtypes: they live in
the checkpoint) match the reference; `forward(sample) -> loss` and `sample(batch_size) -> (B, C, L) CPU
tensor` run on the fused HIP kernels:
    forward: td_q_sample_mixed -> denoiser -> td_mixed_loss (loss and analytic d loss / d model_out)
    sample : td_sample_chain (prior, T x {denoiser, reverse update} as a replayed hipGraph, decode)
Noise comes from the in-kernel Philox generator, seeded from torch's global generator so that
`torch.manual_seed` keeps runs reproducible; parity tests inject the reference's recorded draws through
the `noise=` / `uniforms=` arguments.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .ops import DiffusionKernels, SamplerKernels
from .schedule import make_tables


def _fresh_seed():
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class _LossFn(torch.autograd.Function):
    """loss = mixed_loss(model_out, ...) with the gradient produced by the same fused kernel."""

    @staticmethod
    def forward(ctx, model_out, kernels, noise, x0_idx, xt_idx, t, loss_lambda):
        loss, grad = kernels.mixed_loss(model_out, noise, x0_idx, xt_idx, t, loss_lambda, want_grad=True)
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None, None


class MixedDiffusion(nn.Module):
    def __init__(self, model, seq_length, channels, numerical_features_indices, categorical_features_indices,
                 categorical_num_classes, timesteps=1000, beta_schedule="cosine", auto_normalize=True,
                 loss_lambda=0.5, parametrization="x0"):
        super().__init__()
        if parametrization != "x0":
            raise ValueError("Unknown parametrization: {}".format(parametrization))
        self.auto_normalize = bool(auto_normalize)     # False: the continuous channels are diffused as given (:160-162)
        self.model = model
        self.channels = channels
        self.self_condition = self.model.self_condition
        self.seq_length = seq_length
        self.loss_lambda = loss_lambda
        self.numerical_features_indices = list(numerical_features_indices)
        self.num_numerical_features = len(self.numerical_features_indices)
        self.categorical_features_indices = list(categorical_features_indices)
        self.categorical_num_classes = np.asarray(categorical_num_classes)
        self.parametrization = parametrization

        tables = make_tables(timesteps, beta_schedule)      # raises ValueError on an unknown schedule
        self.num_timesteps = int(timesteps)
        K = [int(k) for k in self.categorical_num_classes]
        expanded = torch.tensor(np.concatenate([np.repeat(k, k) for k in K]) if K else np.zeros(0))
        expanded = expanded.unsqueeze(1).expand(-1, seq_length).to(torch.float32)
        offsets = torch.from_numpy(np.append([0], np.cumsum(K))).long()

        # registration order of mixed_diffusion.py:113-158
        for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "log_alpha", "log_1_min_alpha",
                     "log_cumprod_alpha", "log_1_min_cumprod_alpha"):
            self.register_buffer(name, tables[name])
        self.register_buffer("Lt_history", torch.zeros(timesteps))
        self.register_buffer("Lt_count", torch.zeros(timesteps))
        self.register_buffer("categorical_num_classes_expanded", expanded.contiguous())
        self.register_buffer("offsets", offsets)
        for name in ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
                     "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                     "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2", "loss_weight"):
            self.register_buffer(name, tables[name])
        self._kernels = None
        self._sampler = None

    # -- kernel plumbing -------------------------------------------------------------------------
    def _tables(self):
        return {k: v for k, v in self.named_buffers(recurse=False)}

    def kernels(self) -> DiffusionKernels:
        dev = self.log_alpha.device
        if self._kernels is None or self._kernels.device != dev:
            if dev.type != "cuda":
                raise RuntimeError("MixedDiffusion runs on the MI355X only: move the module to a ROCm device (.cuda())")
            n_cat = len(self.categorical_features_indices)
            self._kernels = DiffusionKernels(
                self.numerical_features_indices, self.categorical_features_indices,
                [int(k) for k in self.categorical_num_classes], self.seq_length, self.num_timesteps, self._tables(),
                # create_samples (:173-179) sizes the result `channels - n_cat`
                self.channels - n_cat, dev, auto_normalize=self.auto_normalize)
            self._sampler = None
            self._num_idx_dev = torch.tensor(self.numerical_features_indices, dtype=torch.int64, device=dev)
        return self._kernels

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, None if k in ("_kernels", "_sampler") else copy.deepcopy(v, memo))
        return new

    # -- training --------------------------------------------------------------------------------
    def forward(self, sample, *args, noise=None, uniforms=None, t=None, **kwargs):
        b, c, n = sample.shape
        assert n == self.seq_length, f'seq length must be {self.seq_length}'
        dk = self.kernels()
        dev = dk.device
        sample = sample.to(dev)
        if t is None:
            t = torch.randint(0, self.num_timesteps, (b,), device=dev).long()
        # a captured training step (ETDiff) reads the seed from device memory: a host draw would be baked into the graph
        seed_dev = getattr(self, "_seed_dev", None)
        x_in, x0_idx, xt_idx, eps = dk.q_sample(sample, t, noise, uniforms, seed=getattr(self, "_seed_base", 0) if seed_dev is not None else _fresh_seed(),
                                                tag=0x40000000, seed_dev=seed_dev)
        # the reference normalises the caller's batch in place (normalize_numericals, :189-194)
        # (a caller that owns the batch and never looks at it again says so: ETDiff's copies of the loader's batches)
        if self.auto_normalize and not getattr(self, "_batch_is_private", False):
            ch = self._num_idx_dev      # the channel list as a device tensor (a list index is a host-to-device copy per call)
            sample[:, ch, :] = sample[:, ch, :] * 2 - 1
        model_out = self.model(x_in, t, None)
        return _LossFn.apply(model_out, dk, eps, x0_idx, xt_idx, t, float(self.loss_lambda))

    # -- sampling --------------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, batch_size, noise=None, uniforms=None, seed=None, patient0=0, to_cpu=True):
        dk = self.kernels()
        den = self.model.kernels()
        if self._sampler is None or self._sampler.den is not den:
            self._sampler = SamplerKernels(den, dk)
            self._film_version = None
        version = self.model._kernels_version
        if self._film_version != version:
            self._sampler.refresh()
            self._film_version = version
        out = self._sampler.sample(batch_size, noise, uniforms, seed=_fresh_seed() if seed is None else seed,
                                   patient0=patient0)
        if to_cpu:
            out = out.cpu()
            _lib.check_async()      # the copy waited for the chain's kernels
        return out
