from timediff_amd.mixed_diffusion import MixedDiffusion  # noqa: F401
