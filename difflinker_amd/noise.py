"""gamma(t) lookup table of the predefined noise schedules (reference ``src/noise.py``).

In scope: ``PredefinedNoiseSchedule`` for ``polynomial_<p>`` (what every released config
uses: ``polynomial_2``, precision 1e-5; train_difflinker.py:139-141) and ``cosine``.
The learned ``GammaNetwork`` is training-only and out of scope.
"""
import numpy as np
import torch


def clip_noise_schedule(alphas2, clip_value=0.001):
    """Clip alpha_t/alpha_{t-1} from below for sampling stability (noise.py:7-19)."""
    alphas2 = np.concatenate([np.ones(1), alphas2], axis=0)
    steps = np.clip(alphas2[1:] / alphas2[:-1], a_min=clip_value, a_max=1.)
    return np.cumprod(steps, axis=0)


def polynomial_schedule(timesteps, s=1e-4, power=3.):
    """alpha^2 of the ``1 - x^power`` schedule (noise.py:22-36)."""
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    alphas2 = clip_noise_schedule((1 - np.power(x / steps, power)) ** 2, clip_value=0.001)
    return (1 - 2 * s) * alphas2 + s


def cosine_beta_schedule(timesteps, s=0.008, raise_to_power=1.):
    """noise.py:39-56."""
    steps = timesteps + 2
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)
    ac = np.cumprod(1. - betas, axis=0)
    return np.power(ac, raise_to_power) if raise_to_power != 1 else ac


class PredefinedNoiseSchedule(torch.nn.Module):
    """Lookup table ``gamma[round(t * timesteps)]`` (noise.py:92-128); the table is a
    non-trainable parameter named ``gamma`` so checkpoints keep the key ``edm.gamma.gamma``."""

    def __init__(self, noise_schedule, timesteps, precision):
        super().__init__()
        self.timesteps = timesteps
        if noise_schedule == 'cosine':
            alphas2 = cosine_beta_schedule(timesteps)
        elif 'polynomial' in noise_schedule:
            splits = noise_schedule.split('_')
            assert len(splits) == 2
            alphas2 = polynomial_schedule(timesteps, s=precision, power=float(splits[1]))
        else:
            raise ValueError(noise_schedule)
        sigmas2 = 1 - alphas2
        gamma = -(np.log(alphas2) - np.log(sigmas2))
        self.gamma = torch.nn.Parameter(torch.from_numpy(gamma).float(), requires_grad=False)

    def forward(self, t):
        return self.gamma[torch.round(t * self.timesteps).long()]
