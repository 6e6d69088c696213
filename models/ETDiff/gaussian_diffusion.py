from timediff_amd.gaussian_diffusion import GaussianDiffusion  # noqa: F401
