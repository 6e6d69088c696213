"""timediff_amd: the MI355X-native ETDiff hot path (denoiser, mixed diffusion, loss, sampler).

Kernels live in timediff_amd/csrc (HIP, gfx950) behind the C ABI of include/timediff_hip.h; the
modules here mirror the reference's Python interfaces on top of it.
"""
__version__ = "0.1.0"
