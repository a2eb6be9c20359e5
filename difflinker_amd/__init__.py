"""difflinker_amd — MI355X-native implementation of DiffLinker's EGNN denoising-diffusion
sampling hot path (``EDM.sample_chain`` -> ``Dynamics.forward`` -> EGNN), behind the reference's
own Python API.  All hot-path arithmetic runs in hand-written HIP kernels for gfx950
(``csrc/``, C ABI in ``include/difflinker_hip.h``); there is no CPU/PyTorch fallback.
"""
from .egnn import Dynamics, DynamicsWithPockets          # noqa: F401
from .edm import EDM, InpaintingEDM                      # noqa: F401
from .lightning import DDPM                              # noqa: F401
from .utils import FoundNaNException                     # noqa: F401

__all__ = ['Dynamics', 'DynamicsWithPockets', 'EDM', 'InpaintingEDM', 'DDPM', 'FoundNaNException']
