"""Flow-match Euler discrete scheduler with dynamic time shifting, as configured for FLUX.1-dev
(diffusers FlowMatchEulerDiscreteScheduler [3p]; call sites
/root/reference/flux_piplines/texturing/pipeline.py:594-610,660).  Host-side scalar logic only;
the per-token update runs in the fused HIP kernel (utx_sched_step)."""
import math

import numpy as np


class FlowMatchEulerConfig:
    num_train_timesteps = 1000
    base_image_seq_len = 256
    max_image_seq_len = 4096
    base_shift = 0.5
    max_shift = 1.15
    use_dynamic_shifting = True


def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15):
    """pipeline.py:59-69 (called with the scheduler-config values at :596-602)."""
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


class FlowMatchEulerScheduler:
    def __init__(self, config=None):
        self.config = config or FlowMatchEulerConfig()
        self.order = 1
        self.sigmas = None
        self.timesteps = None

    def set_timesteps(self, num_inference_steps, mu):
        """sigmas = linspace(1, 1/N, N) (pipeline.py:595) -> exp(mu)/(exp(mu) + (1/s - 1)) ; terminal 0."""
        sig = np.linspace(1.0, 1.0 / num_inference_steps, num_inference_steps).astype(np.float32)
        sig = (math.exp(mu) / (math.exp(mu) + (1.0 / sig - 1.0) ** 1.0)).astype(np.float32)
        self.timesteps = (sig * np.float32(self.config.num_train_timesteps)).astype(np.float32)
        self.sigmas = np.concatenate([sig, np.zeros(1, dtype=np.float32)])
        return self.timesteps

    def dsigma(self, i):
        return float(self.sigmas[i + 1]) - float(self.sigmas[i])
