"""TEST INFRASTRUCTURE ONLY -- ctypes loader for oracle/libcurdle_oracle.so
(the C restatement in oracle/curdle_oracle.c).  Imported by tests/, smoke() and
bench.py's cpu_baseline leg; never by the product package."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_DIR, "libcurdle_oracle.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} not found: run `make -C oracle`")
_lib = C.CDLL(LIB_PATH)
_vp = C.c_void_p
for _n, _a in {
    "oracle_msm_naive": [_vp, _vp, C.c_size_t, _vp],
    "oracle_msm_pippenger": [_vp, _vp, C.c_size_t, C.c_int, C.c_int, _vp],
    "oracle_scalar_mul_gen": [_vp, _vp],
    "oracle_points_walk": [_vp, _vp, C.c_size_t, _vp],
    "oracle_jac_normalise": [_vp, _vp],
    "oracle_fp_mul": [_vp, _vp, _vp],
    "oracle_fr_from_mont": [_vp, _vp],
}.items():
    getattr(_lib, _n).argtypes = _a
    getattr(_lib, _n).restype = C.c_int


def _p(a):
    return a.ctypes.data_as(_vp)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _int_to_limbs(v: int, n: int = 4):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def msm_naive(points, scalars) -> np.ndarray:
    points, scalars = _u64(points), _u64(scalars)
    n = points.shape[0] if points.size else 0
    out = np.zeros(18, dtype=np.uint64)
    _lib.oracle_msm_naive(_p(points), _p(scalars), n, _p(out))
    return out


def msm_pippenger(points, scalars, threads: int = 1, c: int = 0) -> np.ndarray:
    points, scalars = _u64(points), _u64(scalars)
    n = points.shape[0] if points.size else 0
    out = np.zeros(18, dtype=np.uint64)
    _lib.oracle_msm_pippenger(_p(points), _p(scalars), n, threads, c, _p(out))
    return out


def scalar_mul_gen(k: int) -> np.ndarray:
    out = np.zeros(12, dtype=np.uint64)
    _lib.oracle_scalar_mul_gen(_p(_int_to_limbs(k)), _p(out))
    return out


def points_walk(k: int, q: int, n: int) -> np.ndarray:
    out = np.zeros((n, 12), dtype=np.uint64)
    _lib.oracle_points_walk(_p(_int_to_limbs(k)), _p(_int_to_limbs(q)), n, _p(out))
    return out


def jac_normalise(jac) -> np.ndarray:
    jac = _u64(jac)
    out = np.zeros(18, dtype=np.uint64)
    _lib.oracle_jac_normalise(_p(jac), _p(out))
    return out


def fp_mul(a, b) -> np.ndarray:
    out = np.zeros(6, dtype=np.uint64)
    _lib.oracle_fp_mul(_p(_u64(a)), _p(_u64(b)), _p(out))
    return out


def fr_from_mont(a) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint64)
    _lib.oracle_fr_from_mont(_p(_u64(a)), _p(out))
    return out


# ---- the multi-threaded CPU baseline (oracle/cpu_msm_fast.c) ----
def _load_fast(native: bool = False):
    name = "libcurdle_cpufast_native.so" if native else "libcurdle_cpufast.so"
    path = os.path.join(_DIR, name)
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: run `make -C oracle`" + (" native" if native else ""))
    lib = C.CDLL(path)
    lib.fast_msm_g1.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int, _vp]
    lib.fast_msm_g1.restype = C.c_int
    return lib


_fast = {}


def msm_fast(points, scalars, threads: int = 1, c: int = 0, native: bool = False) -> np.ndarray:
    """Bucket-method MSM on `threads` host cores: mulx/adx field products, signed digits, XYZZ
    buckets, every window split over several tasks (oracle/cpu_msm_fast.c)."""
    if native not in _fast:
        _fast[native] = _load_fast(native)
    points, scalars = _u64(points), _u64(scalars)
    n = points.shape[0] if points.size else 0
    out = np.zeros(18, dtype=np.uint64)
    if _fast[native].fast_msm_g1(_p(points), _p(scalars), n, threads, c, _p(out)) != 0:
        raise MemoryError("fast_msm_g1: allocation failed")
    return out
