"""Hot-path helpers of the sampling loop (reference ``src/utils.py``: the six symbols
SURVEY.md section 2 row 4 marks in scope).  Training utilities (EMA, grad-clip queue,
rotation augmentation, stdout tee) are out of scope."""
import random

import numpy as np
import torch


class FoundNaNException(Exception):
    """Raised by ``Dynamics.forward`` when the denoiser output holds a NaN (utils.py:274-289).

    Carries the per-sample index sets the reference computes; callers
    (``generate.py:154-161``) catch it and retry the batch.
    """

    def __init__(self, x=None, h=None, x_nan_idx=None, h_nan_idx=None, f16_range_idx=()):
        x_nan_idx = self.find_nan_idx(x) if x_nan_idx is None else set(x_nan_idx)
        h_nan_idx = self.find_nan_idx(h) if h_nan_idx is None else set(h_nan_idx)
        self.x_h_nan_idx = x_nan_idx & h_nan_idx
        self.only_x_nan_idx = x_nan_idx.difference(h_nan_idx)
        self.only_h_nan_idx = h_nan_idx.difference(x_nan_idx)
        # not in the reference: molecules whose activations left the range of the f16 arithmetic modes (a magnitude bound of
        # 2^75 ~ 3.8e22 or more, e.g. coordinates beyond ~1e10: pack_layout.h, beyond_f16_range).  Their output is void and they
        # are listed under x&h as well, so the reference's callers re-sample them like any NaN; precision='fp32' has no limit.
        self.f16_range_idx = set(f16_range_idx)
        msg = (f'NaN in denoiser output: x&h {sorted(self.x_h_nan_idx)}, '
               f'x only {sorted(self.only_x_nan_idx)}, h only {sorted(self.only_h_nan_idx)}')
        if self.f16_range_idx:
            msg += (f"; of these, {sorted(self.f16_range_idx)} left the range of the f16 arithmetic modes (|value| bound >= 3.8e22): "
                    f"precision='fp32' computes them")
        super().__init__(msg)

    @staticmethod
    def find_nan_idx(z):
        if z is None:
            return set()
        bad = torch.isnan(z).reshape(z.shape[0], -1).any(dim=1)
        return set(torch.nonzero(bad).flatten().tolist())

    @classmethod
    def from_index_sets(cls, x_h, only_x, only_h, f16_range=()):
        """Build from the three index sets themselves (a batch sampled in parts: sets re-numbered to the whole batch)."""
        return cls(x_nan_idx=set(x_h) | set(only_x), h_nan_idx=set(x_h) | set(only_h), f16_range_idx=f16_range)

    @classmethod
    def from_flags(cls, flags):
        """Build from the per-molecule device flag word (bit0: x NaN, bit1: h NaN, bit4: beyond the f16 range - set with both)."""
        flags = flags.tolist() if hasattr(flags, 'tolist') else list(flags)
        return cls(x_nan_idx={i for i, f in enumerate(flags) if f & 1},
                   h_nan_idx={i for i, f in enumerate(flags) if f & 2},
                   f16_range_idx={i for i, f in enumerate(flags) if f & 16})


def set_deterministic(seed):
    """utils.py:263-271."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False


def remove_mean_with_mask(x, node_mask):
    """Subtract the masked mean over atoms (utils.py:56-63)."""
    masked_max_abs_value = (x * (1 - node_mask)).abs().sum().item()
    assert masked_max_abs_value < 1e-5, f'Error {masked_max_abs_value} too high'
    n = node_mask.sum(1, keepdims=True)
    mean = torch.sum(x, dim=1, keepdim=True) / n
    return x - mean * node_mask


def remove_partial_mean_with_mask(x, node_mask, center_of_mass_mask):
    """Subtract the centre of mass of the atoms selected by ``center_of_mass_mask`` from
    every real atom (utils.py:66-74)."""
    x_masked = x * center_of_mass_mask
    n = center_of_mass_mask.sum(1, keepdims=True)
    mean = torch.sum(x_masked, dim=1, keepdim=True) / n
    return x - mean * node_mask


def assert_correctly_masked(variable, node_mask):
    """utils.py:99-101."""
    assert (variable * (1 - node_mask)).abs().max().item() < 1e-4, 'Variables not masked properly.'


def assert_partial_mean_zero_with_mask(x, node_mask, center_of_mass_mask, eps=1e-10):
    """utils.py:90-96."""
    assert_correctly_masked(x, node_mask)
    x_masked = x * center_of_mass_mask
    largest_value = x_masked.abs().max().item()
    error = torch.sum(x_masked, dim=1, keepdim=True).abs().max().item()
    rel_error = error / (largest_value + eps)
    assert rel_error < 1e-2, f'Partial mean is not zero, relative_error {rel_error}'


def sample_gaussian_with_mask(size, device, node_mask):
    """Masked standard normal, drawn from the global generator of ``device`` (utils.py:189-192).

    The draw ORDER of these calls is the noise-seed contract of the sampler."""
    return torch.randn(size, device=device) * node_mask


def split_features(z, n_dims, num_classes, include_charges):
    """utils.py:202-209."""
    assert z.size(2) == n_dims + num_classes + include_charges
    x = z[:, :, 0:n_dims]
    h = {'categorical': z[:, :, n_dims:n_dims + num_classes]}
    if include_charges:
        h['integer'] = z[:, :, n_dims + num_classes:n_dims + num_classes + 1]
    return x, h
