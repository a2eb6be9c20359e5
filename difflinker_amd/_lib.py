"""ctypes binding of ``libdifflinker_hip.so`` (C ABI declared in ``include/difflinker_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` (``hipcc --offload-arch=gfx950``).
There is deliberately NO fallback: if the shared object is missing or cannot be loaded every
compute entry point of the package raises, it never routes to a CPU/PyTorch implementation.
"""
import ctypes
import os

import torch  # noqa: F401  (loads PyTorch-ROCm's libamdhip64 first so both share one HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIFFLINKER_HIP_LIB: load another build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get('DIFFLINKER_HIP_LIB') or os.path.join(_HERE, 'libdifflinker_hip.so')
# the same sources compiled with -DDL_TEST_HOOKS (fault injection for tests/test_gpu_team.py); never loaded by the package itself
TEST_HOOKS_LIB_PATH = os.path.join(_HERE, 'libdifflinker_hip_testhooks.so')

DL_OK = 0
ABI_VERSION = 7
PRECISIONS = {'fp32': 0, 'f16x3': 1, 'f16x2': 2}
DL_ERR_TOO_MANY_ATOMS = -3


class DLConfig(ctypes.Structure):
    _fields_ = [
        ('n_dims', ctypes.c_int32), ('in_node_nf', ctypes.c_int32), ('context_node_nf', ctypes.c_int32),
        ('hidden_nf', ctypes.c_int32), ('n_layers', ctypes.c_int32), ('inv_sublayers', ctypes.c_int32),
        ('condition_time', ctypes.c_int32), ('norm_constant', ctypes.c_float),
        ('normalization_factor', ctypes.c_float), ('precision', ctypes.c_int32),
        ('attention', ctypes.c_int32), ('tanh', ctypes.c_int32), ('coords_range', ctypes.c_float),
        ('aggregation_mean', ctypes.c_int32), ('sin_embedding', ctypes.c_int32),
    ]


class DLSizeConfig(ctypes.Structure):
    _fields_ = [('in_node_nf', ctypes.c_int32), ('hidden_nf', ctypes.c_int32), ('out_node_nf', ctypes.c_int32),
                ('n_layers', ctypes.c_int32)]


class DLInpaintCoef(ctypes.Structure):
    _fields_ = [('alpha_ts', ctypes.c_float), ('c_eps', ctypes.c_float), ('sigma', ctypes.c_float),
                ('a_q', ctypes.c_float), ('b_q', ctypes.c_float), ('decode', ctypes.c_int32),
                ('inv_alpha0', ctypes.c_float), ('sigma0', ctypes.c_float), ('sigma_x', ctypes.c_float),
                ('norm_x', ctypes.c_float), ('norm_h', ctypes.c_float), ('bias_h', ctypes.c_float)]


class DLStepCoef(ctypes.Structure):
    _fields_ = [('t', ctypes.c_float), ('alpha_ts', ctypes.c_float), ('c_eps', ctypes.c_float),
                ('sigma', ctypes.c_float)]


class DLChainArgs(ctypes.Structure):
    _fields_ = [
        ('B', ctypes.c_int32), ('N', ctypes.c_int32), ('T', ctypes.c_int32), ('keep_frames', ctypes.c_int32),
        ('x', ctypes.c_void_p), ('h', ctypes.c_void_p), ('node_mask', ctypes.c_void_p),
        ('fragment_mask', ctypes.c_void_p), ('linker_mask', ctypes.c_void_p), ('edge_mask', ctypes.c_void_p),
        ('context', ctypes.c_void_p), ('noise_x', ctypes.c_void_p), ('noise_h', ctypes.c_void_p),
        ('noise_seed', ctypes.c_uint64), ('mol_offset', ctypes.c_int32), ('team', ctypes.c_int32),
        ('coefs', ctypes.c_void_p),
        ('inv_alpha0', ctypes.c_float), ('sigma0', ctypes.c_float), ('sigma_x', ctypes.c_float),
        ('norm_x', ctypes.c_float), ('norm_h', ctypes.c_float), ('bias_h', ctypes.c_float),
        ('chain', ctypes.c_void_p), ('nan_flags', ctypes.c_void_p), ('nan_step', ctypes.c_void_p),
        ('order', ctypes.c_void_p), ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_size_t),
        ('mol_index', ctypes.c_void_p), ('order_first', ctypes.c_int32), ('order_count', ctypes.c_int32),
        ('q_begin', ctypes.c_void_p), ('q_end', ctypes.c_void_p), ('z_state', ctypes.c_void_p), ('skip_flags', ctypes.c_void_p),
    ]


EXPORTS = ('dl_abi_version', 'dl_last_hip_error', 'dl_max_atoms', 'dl_error_string', 'dl_model_num_tensors',
           'dl_model_create', 'dl_model_destroy', 'dl_egnn_forward_fc', 'dl_sampler_step', 'dl_sample_chain_fc',
           'dl_set_profile_buffer', 'dl_profile_max_events', 'dl_pocket_workspace_bytes', 'dl_egnn_forward_pocket',
           'dl_size_model_num_tensors', 'dl_size_model_create', 'dl_size_model_destroy', 'dl_size_max_fragment_atoms',
           'dl_size_gnn_forward', 'dl_philox_fill', 'dl_egnn_forward_fc_large', 'dl_inpaint_step', 'dl_workspace_bytes',
           'dl_team_max', 'dl_egnn_forward_fc_team', 'dl_team_max_atoms')
TEST_HOOK_EXPORTS = ('dl_debug_team_fault',)       # declared under #ifdef DL_TEST_HOOKS: the test-hooks build only

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises ``HipLibraryError`` when it has not been built."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


class test_hooks:
    """Context manager for the fault-injection tests: inside it ``load()`` returns the ``-DDL_TEST_HOOKS`` build of the
    library (``dl_debug_team_fault`` exists there and nowhere else).  Handles (``dl_model``) are plain structs of device
    pointers and work across the two builds, but the tests create their models inside the context anyway."""

    def __enter__(self):
        global _lib
        self.saved = _lib
        lib = _open(TEST_HOOKS_LIB_PATH)
        lib.dl_debug_team_fault.restype = None
        lib.dl_debug_team_fault.argtypes = [ctypes.c_int32]
        _lib = lib
        return lib

    def __exit__(self, *exc):
        global _lib
        _lib.dl_debug_team_fault(0)
        _lib = self.saved
        return False


def _open(path):
    if not os.path.exists(path):
        raise HipLibraryError(
            f'{path} not found: build the HIP extension first '
            '(python -c "import __graft_entry__ as g; g.build()"). There is no CPU fallback.')
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise HipLibraryError(f'cannot load {path}: {e}') from e
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    lib.dl_abi_version.restype = i32
    lib.dl_last_hip_error.restype = i32
    lib.dl_max_atoms.restype = i32
    lib.dl_error_string.restype = ctypes.c_char_p
    lib.dl_error_string.argtypes = [i32]
    lib.dl_model_num_tensors.restype = i32
    lib.dl_model_num_tensors.argtypes = [ctypes.POINTER(DLConfig)]
    lib.dl_model_create.restype = i32
    lib.dl_model_create.argtypes = [ctypes.POINTER(DLConfig), ctypes.POINTER(vp), i32, ctypes.POINTER(vp)]
    lib.dl_model_destroy.restype = None
    lib.dl_model_destroy.argtypes = [vp]
    lib.dl_egnn_forward_fc.restype = i32
    lib.dl_egnn_forward_fc.argtypes = [vp, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    lib.dl_egnn_forward_fc_team.restype = i32
    lib.dl_egnn_forward_fc_team.argtypes = [vp, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, ctypes.c_size_t, vp]
    lib.dl_workspace_bytes.restype = ctypes.c_size_t
    lib.dl_workspace_bytes.argtypes = [i32, i32]
    lib.dl_team_max.restype = i32
    lib.dl_team_max.argtypes = [i32]
    lib.dl_team_max_atoms.restype = i32
    lib.dl_team_max_atoms.argtypes = [i32]
    lib.dl_sampler_step.restype = i32
    lib.dl_sampler_step.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, DLStepCoef, vp, vp]
    lib.dl_set_profile_buffer.restype = None
    lib.dl_set_profile_buffer.argtypes = [vp]
    lib.dl_profile_max_events.restype = i32
    lib.dl_pocket_workspace_bytes.restype = ctypes.c_size_t
    lib.dl_pocket_workspace_bytes.argtypes = [i32, i32]
    lib.dl_egnn_forward_pocket.restype = i32
    lib.dl_egnn_forward_pocket.argtypes = [vp, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    lib.dl_egnn_forward_fc_large.restype = i32
    lib.dl_egnn_forward_fc_large.argtypes = [vp, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    lib.dl_sample_chain_fc.restype = i32
    lib.dl_sample_chain_fc.argtypes = [vp, ctypes.POINTER(DLChainArgs), vp]
    lib.dl_inpaint_step.restype = i32
    lib.dl_inpaint_step.argtypes = [i32, i32, i32] + [vp] * 10 + [DLInpaintCoef, vp, vp]
    lib.dl_philox_fill.restype = i32
    lib.dl_philox_fill.argtypes = [ctypes.c_uint64, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.dl_size_model_num_tensors.restype = i32
    lib.dl_size_model_num_tensors.argtypes = [ctypes.POINTER(DLSizeConfig)]
    lib.dl_size_model_create.restype = i32
    lib.dl_size_model_create.argtypes = [ctypes.POINTER(DLSizeConfig), ctypes.POINTER(vp), i32, ctypes.POINTER(vp)]
    lib.dl_size_model_destroy.restype = None
    lib.dl_size_model_destroy.argtypes = [vp]
    lib.dl_size_max_fragment_atoms.restype = i32
    lib.dl_size_gnn_forward.restype = i32
    lib.dl_size_gnn_forward.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    return lib


def check(status, what):
    if status != DL_OK:
        lib = load()
        msg = lib.dl_error_string(status).decode()
        raise HipLibraryError(f'{what}: {msg} (status {status}, hip error {lib.dl_last_hip_error()})')


def ptr(t):
    """Device/host pointer of a tensor (or NULL for None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())
