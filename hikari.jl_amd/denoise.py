"""DenoiseConfig (src/denoise.jl:28-54): parameters of the edge-avoiding a-trous wavelet denoiser `denoise!` runs on a Film."""
from . import _abi as A


class DenoiseConfig:
    def __init__(self, iterations=5, sigma_color=4.0, sigma_normal=128.0, sigma_depth=1.0, use_variance=True):
        self.iterations = int(iterations)
        self.sigma_color = float(sigma_color)
        self.sigma_normal = float(sigma_normal)
        self.sigma_depth = float(sigma_depth)
        self.use_variance = bool(use_variance)

    def record(self):
        p = A.hk_denoise_params()
        p.iterations = self.iterations
        p.sigma_color = self.sigma_color
        p.sigma_normal = self.sigma_normal
        p.sigma_depth = self.sigma_depth
        p.use_variance = 1 if self.use_variance else 0
        return p
