"""ctypes binding of `libpromonet_hip.so` (C ABI: include/promonet_hip.h).

The product path has NO CPU fallback: if the library is missing, or a call
is made with CPU tensors, this module raises.
"""
import ctypes
import os
from pathlib import Path

import torch

LIB_PATH = Path(__file__).parent / 'lib' / 'libpromonet_hip.so'

PM_F32, PM_F16, PM_BF16, PM_F16X3, PM_F16A2, PM_F16UX = 0, 1, 2, 3, 4, 5
DTYPES = {'fp32': PM_F32, 'f32': PM_F32, 'f16': PM_F16, 'fp16': PM_F16,
          'bf16': PM_BF16, 'f16x3': PM_F16X3, 'f16a2': PM_F16A2,
          'f16ux': PM_F16UX,
          # FARGAN weight storage only (PM_FARGAN_MIXED, promonet_hip.h):
          # GRU cells / GLU gates f16, the rounding-sensitive layers fp32
          'mixed': 16}
MAX_STAGES, MAX_RESBLOCKS, MAX_DILATIONS = 8, 4, 4
SPARSE_METHODS = {None: 0, 'percentile': 1, 'constant': 2, 'topk': 3}

c_float_p = ctypes.c_void_p
c_int64_p = ctypes.POINTER(ctypes.c_int64)


class HifiganConfig(ctypes.Structure):
    _fields_ = [
        ('num_features', ctypes.c_int),
        ('global_channels', ctypes.c_int),
        ('initial_channels', ctypes.c_int),
        ('num_stages', ctypes.c_int),
        ('upsample_rates', ctypes.c_int * MAX_STAGES),
        ('upsample_kernel_sizes', ctypes.c_int * MAX_STAGES),
        ('num_resblocks', ctypes.c_int),
        ('resblock_kernel_sizes', ctypes.c_int * MAX_RESBLOCKS),
        ('num_dilations', ctypes.c_int),
        ('resblock_dilations', (ctypes.c_int * MAX_DILATIONS) * MAX_RESBLOCKS),
        ('compute_dtype', ctypes.c_int),
        ('stage_compute_dtype', ctypes.c_int * MAX_STAGES)]


# name -> (restype, argtypes); every symbol include/promonet_hip.h declares
_I, _F, _P, _S = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
SIGNATURES = {
    'pm_version': (_I, []),
    'pm_last_error': (ctypes.c_char_p, []),
    'pm_hifigan_create': (_I, [ctypes.POINTER(HifiganConfig),
                               ctypes.POINTER(_P)]),
    'pm_hifigan_destroy': (_I, [_P]),
    'pm_hifigan_load_tensor': (_I, [_P, ctypes.c_char_p, _P, c_int64_p, _I, _P]),
    'pm_hifigan_finalize': (_I, [_P, _P]),
    'pm_hifigan_workspace_bytes': (_S, [_P, _I, _I]),
    'pm_hifigan_hopsize': (_I, [_P]),
    'pm_hifigan_features_cl_channels': (_I, [_P]),
    'pm_hifigan_forward': (_I, [_P, _P, _P, _I, _P, _I, _I, _P, _S, _P]),
    'pm_hifigan_forward_cl': (_I, [_P, _P, _P, _I, _P, _I, _I, _P, _S, _P]),
    'pm_hifigan_forward_ragged': (_I, [_P, _P, _I, _P, _I, _P, _P, _I, _I, _P, _S, _P]),
    'pm_hifigan_profile_enable': (_I, [_P, _I]),
    'pm_hifigan_profile_only': (_I, [_P, ctypes.c_char_p]),
    'pm_hifigan_profile_collect': (_I, [_P]),
    'pm_hifigan_profile_reset': (_I, [_P]),
    'pm_hifigan_profile_report': (ctypes.c_char_p, [_P]),
    'pm_prepare_features': (_I, [_P] * 8 + [_I] * 9 + [_F] * 6 + [_P]),
    'pm_prepare_global_features': (_I, [_P] * 5 + [_I, _I, _I, _P]),
    'pm_prepare_global_features_linear': (_I, [_P] * 6 + [_I, _I, _I, _P]),
    'pm_op_workspace_bytes': (_S, [_I, _I, _I]),
    'pm_block_iteration_cl': (_I, [_I] + [_P] * 6 + [_I] * 6 + [_F, _P, _S, _P]),
    'pm_block_cl': (_I, [_I, _P, _P] + [_P] * 5 + [_I] * 6 + [_F, _P, _S, _P]),
    'pm_block_act16_cl': (
        _I, [_I, _I, _P, _P, _P] + [_P] * 5 + [_I] * 6 + [_F, _P, _S, _P]),
    'pm_mrf_cl': (_I, [_I, _P, _P] + [_P] * 5 + [_I] * 4 + [_P, _S, _P]),
    'pm_input_conv_cl': (_I, [_I] + [_P] * 7 + [_I] * 6 + [_P, _S, _P]),
    'pm_conv_transpose_cl': (_I, [_I] + [_P] * 4 + [_I] * 6 + [_P, _S, _P]),
    'pm_conv_transpose_x16_cl': (_I, [_I] + [_P] * 4 + [_I] * 5 + [_P, _S, _P]),
    'pm_out_conv_tanh': (_I, [_P, _P, _P, _I, _I, _I, _P]),
    'pm_debug_timeline': (_I, [_P]),
    'pm_debug_force': (_I, [_I, _I]),
    'pm_debug_skew': (_I, [_I]),
    'pm_mfma_probe': (_I, [_I, _I, _P, _P, _I, _P]),
    'pm_walk_scratch_bytes': (_S, [_I]),
    'pm_fold_weight_norm': (_I, [_P, _P, _P, _I, _I, _P]),
    'pm_to_channels_last': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'pm_grid_sample': (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _F, _P]),
    'pm_stretch_grid': (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _P]),
    'pm_fargan_create': (_I, [_I, _I, _I, ctypes.POINTER(_P)]),
    'pm_fargan_destroy': (_I, [_P]),
    'pm_fargan_load_tensor': (_I, [_P, ctypes.c_char_p, _P, c_int64_p, _I, _P]),
    'pm_fargan_finalize': (_I, [_P, _P]),
    'pm_fargan_workspace_bytes': (_S, [_P, _I, _I]),
    'pm_fargan_forward': (_I, [_P, _P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _S, _P]),
    'pm_fargan_forward_ragged': (_I, [_P, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _P, _S, _P]),
    'pm_fargan_set_mode': (_I, [_P, _I]),
    'pm_fargan_check': (_I, [_P, _I, _I, _P, _P]),
    'pm_stft_scratch_bytes': (_S, [_I, _I]),
    'pm_stft_magnitude': (_I, [_P, _P, _I, _I, _P, _S, _P]),
    'pm_stft_magnitude_dft': (_I, [_P, _P, _I, _I, _P, _S, _P]),
    'pm_stft_mel_scratch_bytes': (_S, [_I]),
    'pm_stft_mel_prepare': (_I, [_P, _I, _P, _S, _P]),
    'pm_stft_mel': (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P]),
    'pm_stft_set_frames_per_group': (_I, [_I]),
    'pm_stft_set_loudness_passes': (_I, [_I]),
    'pm_stft_launch_info': (_I, [_I, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    'pm_linear_to_mel': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    'pm_stft_backward_scratch_bytes': (_S, [_I, _I]),
    'pm_stft_magnitude_backward': (_I, [_P, _P, _P, _I, _I, _P, _S, _P]),
    'pm_linear_to_mel_backward': (_I, [_P] * 5 + [_I] * 5 + [_F, _P]),
    'pm_loudness_scratch_bytes': (_S, [_I, _I]),
    'pm_loudness': (_I, [_P, _P, _P, _I, _I, _I, _F, _P, _S, _P]),
}

_lib = None


def lib():
    """Load (once) and return the shared library; raise if it is missing."""
    global _lib
    if _lib is None:
        path = Path(os.environ.get('PROMONET_HIP_LIB', LIB_PATH))
        if not path.exists():
            raise RuntimeError(
                f'{path} not found: build it with `make -j4` (or '
                '`python -c "import __graft_entry__ as g; g.build()"`). '
                'promonet_amd has no CPU fallback.')
        handle = ctypes.CDLL(str(path))
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


PM_OK, PM_EINVAL, PM_ESTATE, PM_EHIP, PM_ENOMEM = 0, -1, -2, -3, -4
PM_ETIMEOUT = -5


class LibraryError(RuntimeError):
    """A non-zero status of the C ABI; `.code` is the PM_E* value."""

    def __init__(self, code, message):
        super().__init__(f'libpromonet_hip error {code}: {message}')
        self.code = code


def check(code):
    if code != 0:
        raise LibraryError(code, lib().pm_last_error().decode())


def ptr(tensor, dtype=torch.float32):
    """Device pointer of a contiguous CUDA/HIP tensor."""
    if tensor is None:
        return None
    if not tensor.is_cuda:
        raise RuntimeError(
            'promonet_amd runs on an AMD GPU only (tensor is on '
            f'{tensor.device}); there is no CPU fallback')
    if tensor.dtype != dtype:
        raise RuntimeError(f'expected {dtype}, got {tensor.dtype}')
    if not tensor.is_contiguous():
        raise RuntimeError('tensor must be contiguous')
    return tensor.data_ptr()


def require_gpu(tensor):
    if not tensor.is_cuda:
        raise RuntimeError(
            'promonet_amd runs on an AMD GPU only (tensor is on '
            f'{tensor.device}); there is no CPU fallback')


def stream():
    return torch.cuda.current_stream().cuda_stream


def shape_array(shape):
    return (ctypes.c_int64 * len(shape))(*shape)
