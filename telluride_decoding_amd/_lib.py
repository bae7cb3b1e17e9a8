"""ctypes binding of libtd_hotpath.so (the C-ABI in include/td_hotpath.h).

The product path has no CPU fallback: if the HIP library is missing or no
MI355X is visible, every entry point raises `HotPathUnavailable`.
"""
import ctypes
import os
import re

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
# (TD_HOTPATH_LIB: development -- A/B runs of two builds of the library in one GPU session)
LIB_PATH = os.environ.get('TD_HOTPATH_LIB') or os.path.join(PKG, 'libtd_hotpath.so')
HEADER = os.path.join(os.path.dirname(PKG), 'include', 'td_hotpath.h')

TD_OK = 0
TD_ERR_INVALID = -1
TD_ERR_HIP = -2
TD_ERR_SINGULAR = -3
TD_ERR_NOMEM = -4
TD_ERR_STATE = -5
TD_SOLVER_AUTO, TD_SOLVER_CHOLESKY, TD_SOLVER_CG = 0, 1, 2


class HotPathUnavailable(RuntimeError):
  """The HIP extension (or a GPU) is missing; there is no CPU fallback."""


class HotPathError(RuntimeError):
  pass


_c = ctypes
_vp, _i, _i64, _f, _d, _sz = (_c.c_void_p, _c.c_int, _c.c_int64, _c.c_float,
                              _c.c_double, _c.c_size_t)
_pi64 = _c.POINTER(_c.c_int64)
_pd = _c.POINTER(_c.c_double)

# name -> argtypes (restype is int unless listed in _RESTYPE)
SIGNATURES = {
    'td_version': [],
    'td_device_count': [_c.POINTER(_i)],
    'td_create': [_i, _c.POINTER(_vp)],
    'td_destroy': [_vp],
    'td_last_error': [_vp],
    'td_set_stream': [_vp, _vp],
    'td_use_own_stream': [_vp],
    'td_stream_create_masked': [_i, _i, _i, _c.POINTER(_vp)],
    'td_stream_destroy': [_vp],
    'td_set_accumulate_mode': [_vp, _i],
    'td_synchronize': [_vp],
    'td_malloc': [_vp, _sz, _c.POINTER(_vp)],
    'td_free': [_vp, _vp],
    'td_memcpy_h2d': [_vp, _vp, _vp, _sz],
    'td_memcpy_d2h': [_vp, _vp, _vp, _sz],
    'td_memset': [_vp, _vp, _i, _sz],
    'td_timer_start': [_vp],
    'td_timer_stop': [_vp, _c.POINTER(_f)],
    'td_profile_enable': [_vp, _i],
    'td_set_cu_count': [_vp, _i],
    'td_probe_bf16_mfma': [_vp, _i, _pd],
    'td_profile_read': [_vp, _pi64, _pd, _pd],
    'td_stats_create': [_vp, _i, _i, _i, _i, _i, _i, _i, _c.POINTER(_vp)],
    'td_stats_destroy': [_vp, _vp],
    'td_stats_reset': [_vp, _vp],
    'td_stats_complete': [_vp, _vp],
    'td_stats_accumulate': [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _pi64, _i, _i,
                            _pi64],
    'td_stats_accumulate_parts': [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _pi64, _i, _i,
                            _pi64, _i],
    'td_stats_accumulate_ranges': [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _pi64, _i, _i,
                                   _pi64, _pi64, _pi64, _c.POINTER(_i), _i],
    'td_stats_counts': [_vp, _vp, _pi64, _pi64],
    'td_stats_combine': [_vp, _vp, _c.POINTER(_vp), _i],
    'td_stats_packed_len': [_vp, _vp, _i64, _pi64],
    'td_stats_pack': [_vp, _vp, _vp, _i64, _i64],
    'td_stats_unpack': [_vp, _vp, _vp, _i64],
    'td_stats_unpack_known': [_vp, _vp, _vp, _i64, _i64],
    'td_stats_accumulate_each': [_vp, _c.POINTER(_vp), _vp, _i64, _vp, _i64, _pi64, _i, _i, _pi64, _c.POINTER(_i)],
    'td_stats_allreduce': [_vp, _vp, _vp, _i64, _i64, _i64],
    'td_allreduce_f64': [_vp, _vp, _i64, _vp],
    'td_rccl_available': [_vp],
    'td_rccl_unique_id': [_vp, _vp],
    'td_rccl_comm_create': [_vp, _i, _i, _vp, _c.POINTER(_vp)],
    'td_rccl_comm_count': [_vp, _vp, _c.POINTER(_i)],
    'td_rccl_comm_destroy': [_vp, _vp],
    'td_stats_moments': [_vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'td_ridge_solve': [_vp, _vp, _pd, _i, _vp, _vp],
    'td_set_solver': [_vp, _i],
    'td_set_option': [_vp, _c.c_char_p, _i64],
    'td_last_solve_info': [_vp, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i)],
    'td_ridge_solve_async': [_vp, _vp, _pd, _i, _vp, _vp, _vp],
    'td_ridge_solve_multi': [_vp, _c.POINTER(_vp), _i, _pd, _i, _vp, _vp, _vp],
    'td_ridge_solve_loso': [_vp, _vp, _c.POINTER(_vp), _i, _pd, _i, _i, _d, _vp, _vp, _c.POINTER(_i),
                            _c.POINTER(_i)],
    'td_ridge_solve_loso_terms': [_vp, _vp, _c.POINTER(_vp), _c.POINTER(_i), _pd, _i, _pd, _i, _i, _d, _i, _vp, _vp,
                                  _c.POINTER(_i), _c.POINTER(_i)],
    'td_spd_solve': [_vp, _vp, _vp, _i, _i, _i],
    'td_general_solve': [_vp, _vp, _vp, _i, _i],
    'td_shrinkage_moment': [_vp, _vp, _i64, _i, _i, _i, _pi64, _i, _i, _pi64, _i64, _vp],
    'td_shrinkage_terms': [_vp, _vp, _i64, _i, _vp, _c.c_double, _pd],
    'td_shrunk_covariance': [_vp, _vp, _i64, _i, _c.c_double, _c.c_double, _vp, _vp],
    'td_predict_fir': [_vp, _vp, _i64, _pi64, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i64],
    'td_predict_fir_per_file': [_vp, _vp, _i64, _pi64, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i64],
    'td_cca_transform': [_vp, _vp, _i64, _i, _i, _i, _vp, _i64, _i, _i, _i, _pi64, _i, _i,
                         _vp, _vp, _vp, _vp, _i, _vp, _i64],
    'td_cca_solve': [_vp, _vp, _d, _d, _d, _i, _vp, _vp, _vp, _vp, _vp, _c.POINTER(_i)],
    'td_sym_eigh': [_vp, _vp, _i, _vp, _vp, _c.POINTER(_i)],
    'td_jacobi_svd': [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _c.POINTER(_i)],
    'td_window_count': [_pi64, _i, _i, _i, _pi64, _pi64],
    'td_window_sums': [_vp, _vp, _i64, _vp, _i64, _i, _pi64, _i, _i, _i, _vp],
    'td_window_sums_cycled': [_vp, _vp, _i64, _vp, _i64, _i, _i, _pi64, _i, _i, _i, _vp],
    'td_window_scores': [_vp, _vp, _i64, _i, _i, _i, _i, _pd, _pd, _pd, _vp],
    'td_window_pearson': [_vp, _vp, _i64, _i, _i, _i, _vp],
    'td_frame_scores': [_vp, _vp, _i64, _vp, _i64, _i, _i64, _i, _pd, _pd, _pd, _pd, _d,
                        _d, _vp],
    'td_window_means': [_vp, _vp, _pi64, _i, _i, _i, _vp],
    'td_decide_wta': [_vp, _vp, _vp, _i64, _vp],
    'td_decide_step': [_vp, _vp, _vp, _pi64, _i, _vp, _pd],
    'td_decode_ssd': [_vp, _vp, _vp, _pi64, _i, _pd, _pd, _vp],
    'td_decode_ssd_stream': [_vp, _vp, _vp, _pi64, _i, _pd, _pd, _vp, _vp],
    'td_ssd_state_doubles': [],
    'td_decode_fused': [_vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _i64, _pi64, _i, _i,
                        _i, _pd, _vp, _vp],
}
_RESTYPE = {'td_last_error': _c.c_char_p}

_lib = None


def header_symbols():
  """Every function name the C header declares."""
  with open(HEADER) as f:
    text = f.read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(td_[a-z0-9_]+)\s*\(', text)))


def load():
  """Loads the shared library (no GPU needed) and sets the prototypes."""
  global _lib
  if _lib is not None:
    return _lib
  try:
    # PyTorch-ROCm ships its own libamdhip64; it must be the one HIP runtime of the
    # process.  Loading ours first (it resolves libamdhip64 from /opt/rocm) and torch
    # afterwards leaves torch without devices ("No HIP GPUs are available").
    import torch  # noqa: F401
  except ImportError:
    pass
  if not os.path.exists(LIB_PATH):
    raise HotPathUnavailable(
        'HIP extension %s is missing: build it with '
        '`python -m telluride_decoding_amd.build` (hipcc, gfx950). '
        'There is no CPU fallback for the hot path.' % LIB_PATH)
  lib = ctypes.CDLL(LIB_PATH)
  for name, args in SIGNATURES.items():
    fn = getattr(lib, name)   # AttributeError if the .so lacks a declared symbol
    fn.argtypes = args
    fn.restype = _RESTYPE.get(name, _i)
  _lib = lib
  return lib


def check(handle_ptr, status):
  if status == TD_OK:
    return
  lib = load()
  msg = lib.td_last_error(handle_ptr)
  msg = msg.decode() if msg else 'status %d' % status
  if status == TD_ERR_INVALID:
    raise ValueError(msg)
  if status == TD_ERR_SINGULAR:
    raise np.linalg.LinAlgError(msg)
  if status == TD_ERR_NOMEM:
    raise MemoryError(msg)
  if status == TD_ERR_HIP and handle_ptr is None:
    raise HotPathUnavailable(msg)
  raise HotPathError(msg)


def i64_array(values):
  arr = np.ascontiguousarray(values, dtype=np.int64)
  return arr, arr.ctypes.data_as(_pi64)


def f64_array(values):
  arr = np.ascontiguousarray(values, dtype=np.float64)
  return arr, arr.ctypes.data_as(_pd)
