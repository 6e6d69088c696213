"""GaussianDiffusion with the reference's interface (models/ETDiff/gaussian_diffusion.py:25-190): the
continuous-only variant (`--diff_type gaussian`), i.e. the sum K = 0 case of the same fused kernels.
Buffers are registered with the reference's names and order (gaussian_diffusion.py:50-72)."""
import torch
import torch.nn as nn

from .mixed_diffusion import _LossFn, _fresh_seed
from .ops import DiffusionKernels, SamplerKernels
from .schedule import make_tables

_GAUSS_BUFFERS = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                  "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                  "posterior_mean_coef1", "posterior_mean_coef2", "loss_weight")


class GaussianDiffusion(nn.Module):
    def __init__(self, model, seq_length, channels, timesteps=1000, beta_schedule="cosine", auto_normalize=True):
        super().__init__()
        self.auto_normalize = bool(auto_normalize)     # False: the data is diffused as given (gaussian_diffusion.py:95-96)
        self.model = model
        self.channels = channels
        self.self_condition = self.model.self_condition
        self.seq_length = seq_length
        self._all_tables = make_tables(timesteps, beta_schedule)
        self.num_timesteps = int(timesteps)
        for name in _GAUSS_BUFFERS:
            self.register_buffer(name, self._all_tables[name])
        self._kernels = None
        self._sampler = None
        self._film_version = None

    def kernels(self) -> DiffusionKernels:
        dev = self.betas.device
        if self._kernels is None or self._kernels.device != dev:
            if dev.type != "cuda":
                raise RuntimeError("GaussianDiffusion runs on the MI355X only: move the module to a ROCm device")
            self._kernels = DiffusionKernels(list(range(self.channels)), [], [], self.seq_length, self.num_timesteps,
                                             self._all_tables, self.channels, dev, auto_normalize=self.auto_normalize)
            self._sampler = None
        return self._kernels

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, None if k in ("_kernels", "_sampler") else copy.deepcopy(v, memo))
        return new

    def forward(self, sample, *args, noise=None, t=None, **kwargs):
        b, c, n = sample.shape
        assert n == self.seq_length, f'seq length must be {self.seq_length}'
        dk = self.kernels()
        sample = sample.to(dk.device)
        if t is None:
            t = torch.randint(0, self.num_timesteps, (b,), device=dk.device).long()
        seed_dev = getattr(self, "_seed_dev", None)      # set by ETDiff while it captures / replays a training step
        x_in, x0_idx, xt_idx, eps = dk.q_sample(sample, t, noise, None, seed=getattr(self, "_seed_base", 0) if seed_dev is not None else _fresh_seed(),
                                                tag=0x40000000, seed_dev=seed_dev)
        model_out = self.model(x_in, t, None)
        return _LossFn.apply(model_out, dk, eps, x0_idx, xt_idx, t, 0.0)

    @torch.no_grad()
    def sample(self, batch_size=16, noise=None, seed=None, patient0=0):
        """-> (B, channels, L) on the device (the reference's Gaussian sampler does not move to the CPU)."""
        dk = self.kernels()
        den = self.model.kernels()
        if self._sampler is None or self._sampler.den is not den:
            self._sampler = SamplerKernels(den, dk)
            self._film_version = None
        if self._film_version != self.model._kernels_version:
            self._sampler.refresh()
            self._film_version = self.model._kernels_version
        return self._sampler.sample(batch_size, noise, None, seed=_fresh_seed() if seed is None else seed,
                                    patient0=patient0)
