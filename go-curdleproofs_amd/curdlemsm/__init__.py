"""ctypes binding of libcurdlemsm.so (include/curdle_msm.h).

Thin plumbing only: every function below forwards to one C-ABI entry point of
the HIP library; there is no Python or CPU implementation of the MSM here.  If
the shared library is missing, importing this package raises -- the product
path must fail loudly rather than fall back.

Layouts are gnark-crypto's (see the header): points are uint64[n, 12]
(G1Affine, Montgomery), scalars uint64[n, 4] (fr.Element, Montgomery), results
uint64[18] (G1Jac, canonical representative).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CURDLE_MSM_LIB", os.path.join(os.path.dirname(_HERE), "libcurdlemsm.so"))

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C go-curdleproofs_amd` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`)")

# PyTorch-ROCm bundles its own libamdhip64; two HIP runtimes in one process do not
# both see the GPU.  When torch is present (it provides device memory, streams and
# torch.distributed to callers of this binding) load it first so that
# libcurdlemsm.so binds to the runtime already in the process.
try:  # pragma: no cover - depends on the environment
    import torch  # noqa: F401
except Exception:  # torch absent: the system ROCm runtime is used
    torch = None

_lib = C.CDLL(LIB_PATH)

OK, EINVAL, ENODEV, EHIP, ENOMEM, EBUSY = 0, -1, -2, -3, -4, -5
MSM_SLOTS = 8

# Every symbol include/curdle_msm.h declares (tests check they are all exported).
SYMBOLS = [
    "curdle_init", "curdle_shutdown", "curdle_last_error", "curdle_plan_override", "curdle_device_available",
    "curdle_init_devices", "curdle_device_count", "curdle_set_device", "curdle_get_device", "curdle_get_device_selection", "curdle_stat_spread_calls",
    "curdle_msm_g1_replicated",
    "curdle_msm_g1_ex", "curdle_msm_g1_device_ex", "curdle_msm_g1_device_windows_ex", "curdle_msm_g1_device_submit_ex",
    "curdle_msm_forget_bases",
    "curdle_msm_g1", "curdle_msm_g1_device", "curdle_msm_g1_device_windows",
    "curdle_msm_g1_device_submit", "curdle_msm_wait",
    "curdle_msm_window_bits", "curdle_msm_num_windows", "curdle_msm_window_widths", "curdle_msm_num_windows_ex",
    "curdle_msm_window_widths_ex", "curdle_g1_sum",
    "curdle_msm_g1_batch", "curdle_msm_g1_batch_device", "curdle_msm_g1_multi",
    "curdle_rand_new", "curdle_rand_free", "curdle_rand_get_fr", "curdle_rand_get_g1_affine",
    "curdle_rand_permutation",
    "curdle_acc_new", "curdle_acc_free", "curdle_acc_accumulate_check", "curdle_acc_accumulate_check_deferred",
    "curdle_acc_verify",
    "curdle_acc_get_A_c", "curdle_acc_num_bases", "curdle_acc_export",
    "curdle_profile_enable", "curdle_profile_last", "curdle_selftest_op", "curdle_selftest_shape", "curdle_msm_free_slots",
    "curdle_synth_points_walk_device",
    "curdle_crs_generate", "curdle_crs_free", "curdle_crs_size", "curdle_shuffle_permute_commit",
    "curdle_prove", "curdle_verify", "curdle_proof_from_bytes", "curdle_proof_free", "curdle_verify_proof",
    "curdle_verify_batch", "curdle_verify_set_eager",
    "curdle_whisk_is_valid_shuffle_proof", "curdle_whisk_is_valid_shuffle_proof_batch",
    "curdle_whisk_generate_shuffle_proof",
    "curdle_whisk_is_valid_tracker_proof", "curdle_whisk_generate_tracker_proof", "curdle_proof_reencode", "curdle_merlin_test_vector", "curdle_g1_decompress_batch", "curdle_g1_decompress_begin", "curdle_g1_decompress_finish", "curdle_g1_decompress_start", "curdle_g1_decompress_points",
    "curdle_g1_scalar_mul_batch",
    "curdle_g1_compress", "curdle_g1_decompress", "curdle_set_last_error", "curdle_fr_inner_product",
    "curdle_dbases_create", "curdle_dbases_free", "curdle_dbases_size", "curdle_dbases_valid",
    "curdle_msm_g1_dbases", "curdle_msm_g1_dbases_host", "curdle_msm_g1_dbases_windows", "curdle_msm_g1_dbases_submit",
    "curdle_dacc_begin", "curdle_dacc_run", "curdle_dacc_submit", "curdle_dacc_poll", "curdle_dacc_wait", "curdle_dacc_abort",
    "curdle_verify_set_device_acc", "curdle_verify_export_accumulator",
]

_u64p = C.POINTER(C.c_uint64)
_vp = C.c_void_p


class _Profile(C.Structure):
    _fields_ = [("n_kernels", C.c_int), ("name", C.c_char_p * 16), ("ms", C.c_float * 16),
                ("window_bits", C.c_int), ("num_windows", C.c_int), ("entries", C.c_ulonglong),
                ("fragments", C.c_ulonglong)]


def _sig(name, restype, *argtypes):
    f = getattr(_lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_init = _sig("curdle_init", C.c_int, C.c_int)
_shutdown = _sig("curdle_shutdown", C.c_int)
_init_devices = _sig("curdle_init_devices", C.c_int, C.POINTER(C.c_int), C.c_int)
_device_count = _sig("curdle_device_count", C.c_int)
_set_device = _sig("curdle_set_device", C.c_int, C.c_int)
_get_device = _sig("curdle_get_device", C.c_int)
_get_device_selection = _sig("curdle_get_device_selection", C.c_int)
_stat_spread_calls = _sig("curdle_stat_spread_calls", C.c_ulonglong)
_msm_g1_ex = _sig("curdle_msm_g1_ex", C.c_int, _vp, _vp, C.c_size_t, C.c_uint, _vp)
_msm_g1_device_windows_ex = _sig("curdle_msm_g1_device_windows_ex", C.c_int, _vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                 C.c_uint, _vp, _vp)
_msm_submit_ex = _sig("curdle_msm_g1_device_submit_ex", C.c_int, _vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint,
                      C.POINTER(C.c_int))
_msm_forget_bases = _sig("curdle_msm_forget_bases", C.c_int, _vp)
_msm_g1_replicated = _sig("curdle_msm_g1_replicated", C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t,
                          C.c_int, _vp)
_last_error = _sig("curdle_last_error", C.c_int, C.c_char_p, C.c_size_t)
_plan_override = _sig("curdle_plan_override", C.c_int, C.c_char_p, C.c_longlong)
_device_available = _sig("curdle_device_available", C.c_int)
_msm_g1 = _sig("curdle_msm_g1", C.c_int, _vp, _vp, C.c_size_t, _vp)
_msm_g1_device = _sig("curdle_msm_g1_device", C.c_int, _vp, _vp, C.c_size_t, _vp, _vp)
_msm_g1_device_windows = _sig("curdle_msm_g1_device_windows", C.c_int, _vp, _vp, C.c_size_t, C.c_int, C.c_int,
                              C.c_int, _vp, _vp)
_msm_submit = _sig("curdle_msm_g1_device_submit", C.c_int, _vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int,
                   C.POINTER(C.c_int))
_msm_wait = _sig("curdle_msm_wait", C.c_int, C.c_int, _vp)
_window_bits = _sig("curdle_msm_window_bits", C.c_int, C.c_size_t)
_num_windows = _sig("curdle_msm_num_windows", C.c_int, C.c_size_t, C.c_int)
_window_widths = _sig("curdle_msm_window_widths", C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_int))
_num_windows_ex = _sig("curdle_msm_num_windows_ex", C.c_int, C.c_size_t, C.c_int, C.c_uint)
_window_widths_ex = _sig("curdle_msm_window_widths_ex", C.c_int, C.c_size_t, C.c_int, C.c_uint, C.POINTER(C.c_int))
_msm_batch_device = _sig("curdle_msm_g1_batch_device", C.c_int, _vp, _vp, _vp, C.c_size_t, _vp, _vp)
_g1_sum = _sig("curdle_g1_sum", C.c_int, _vp, C.c_size_t, _vp)
_msm_batch = _sig("curdle_msm_g1_batch", C.c_int, _vp, _vp, _vp, C.c_size_t, _vp)
_msm_multi = _sig("curdle_msm_g1_multi", C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp)
_rand_new = _sig("curdle_rand_new", _vp, C.c_uint64)
_rand_free = _sig("curdle_rand_free", None, _vp)
_rand_get_fr = _sig("curdle_rand_get_fr", C.c_int, _vp, _vp)
_rand_get_g1 = _sig("curdle_rand_get_g1_affine", C.c_int, _vp, _vp)
_rand_perm = _sig("curdle_rand_permutation", C.c_int, _vp, C.c_size_t, _vp)
_acc_new = _sig("curdle_acc_new", _vp)
_acc_free = _sig("curdle_acc_free", None, _vp)
_acc_check = _sig("curdle_acc_accumulate_check", C.c_int, _vp, _vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp)
_acc_check_deferred = _sig("curdle_acc_accumulate_check_deferred", C.c_int, _vp, _vp, _vp, C.c_size_t, _vp, C.c_size_t,
                           _vp, C.c_size_t, _vp)
_acc_verify = _sig("curdle_acc_verify", C.c_int, _vp, C.POINTER(C.c_int))
_acc_get_A_c = _sig("curdle_acc_get_A_c", C.c_int, _vp, _vp)
_acc_num_bases = _sig("curdle_acc_num_bases", C.c_size_t, _vp)
_acc_export = _sig("curdle_acc_export", C.c_int, _vp, _vp, _vp)
_profile_enable = _sig("curdle_profile_enable", C.c_int, C.c_int)
_profile_last = _sig("curdle_profile_last", C.c_int, C.POINTER(_Profile))
_synth_walk = _sig("curdle_synth_points_walk_device", C.c_int, _vp, _vp, C.c_size_t, _vp)
_dbases_create = _sig("curdle_dbases_create", C.c_int, _vp, C.c_size_t, C.POINTER(C.c_void_p))
_dbases_free = _sig("curdle_dbases_free", None, _vp)
_msm_dbases_windows = _sig("curdle_msm_g1_dbases_windows", C.c_int, _vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int, _vp)
_msm_dbases_host = _sig("curdle_msm_g1_dbases_host", C.c_int, _vp, _vp, C.c_size_t, _vp)
_msm_dbases_submit = _sig("curdle_msm_g1_dbases_submit", C.c_int, _vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int))
_selftest_op = _sig("curdle_selftest_op", C.c_int, C.c_int, _vp, C.c_size_t, _vp, C.c_int)
_selftest_shape = _sig("curdle_selftest_shape", C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))


_crs_generate = _sig("curdle_crs_generate", _vp, C.c_size_t, _vp)
_crs_free = _sig("curdle_crs_free", None, _vp)
_crs_size = _sig("curdle_crs_size", C.c_size_t, _vp)
_spc = _sig("curdle_shuffle_permute_commit", C.c_int, _vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, _vp)
_prove = _sig("curdle_prove", C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t,
              C.POINTER(C.c_size_t))
_verify = _sig("curdle_verify", C.c_int, _vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp,
               C.POINTER(C.c_int))
_proof_from_bytes = _sig("curdle_proof_from_bytes", C.c_int, _vp, C.c_size_t, C.POINTER(_vp))
_proof_free = _sig("curdle_proof_free", None, _vp)
_verify_proof = _sig("curdle_verify_proof", C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp, C.POINTER(C.c_int))
_verify_batch = _sig("curdle_verify_batch", C.c_int, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp,
                     C.c_int, _vp)
_whisk_valid_shuffle = _sig("curdle_whisk_is_valid_shuffle_proof", C.c_int, _vp, _vp, _vp, C.c_size_t, C.c_size_t, _vp, _vp,
                            C.POINTER(C.c_int))
_whisk_valid_shuffle_batch = _sig("curdle_whisk_is_valid_shuffle_proof_batch", C.c_int, _vp, C.c_size_t, _vp, _vp, C.c_size_t,
                                  _vp, _vp, C.c_int, _vp)
_whisk_gen_shuffle = _sig("curdle_whisk_generate_shuffle_proof", C.c_int, _vp, _vp, C.c_size_t, _vp, _vp, _vp)
_whisk_valid_tracker = _sig("curdle_whisk_is_valid_tracker_proof", C.c_int, _vp, _vp, _vp, C.POINTER(C.c_int))
_whisk_gen_tracker = _sig("curdle_whisk_generate_tracker_proof", C.c_int, _vp, _vp, _vp, _vp)
_verify_set_eager = _sig("curdle_verify_set_eager", C.c_int, C.c_int)
_reencode = _sig("curdle_proof_reencode", C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_size_t))
_merlin_tv = _sig("curdle_merlin_test_vector", C.c_int, C.c_char_p, C.c_char_p, _vp, C.c_size_t, C.c_char_p, _vp,
                  C.c_size_t)
_g1_compress = _sig("curdle_g1_compress", C.c_int, _vp, _vp)
_g1_decompress = _sig("curdle_g1_decompress", C.c_int, _vp, C.c_int, _vp)


class CurdleError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"curdle error {code}: {msg}")
        self.code = code
        self.msg = msg


def last_error() -> str:
    buf = C.create_string_buffer(256)
    _last_error(buf, 256)
    return buf.value.decode()


def _check(rc: int) -> None:
    if rc != OK:
        raise CurdleError(rc, last_error())


def _as_u64(a, cols=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if cols is not None and a.size and a.shape[-1] != cols:
        raise ValueError(f"expected last dimension {cols}, got {a.shape}")
    return a


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_vp)


def plan_override(name: str, value=None) -> None:
    """Test / measurement hook: set one of the library's knobs (host/knobs.h; the name with or without
    CURDLE_), or put it back to "not set" with value None.  The environment is read only once."""
    _check(_plan_override(name.encode(), -1 if value is None else int(value)))


class knobs:
    """with knobs(WINDOW_BITS=12, SEG_LEN=16): ... -- the knobs are unset again on the way out."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        for k, v in self.kv.items():
            plan_override(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kv:
            plan_override(k, None)
        return False


def init(device: int = 0) -> None:
    _check(_init(device))


def shutdown() -> None:
    _check(_shutdown())


def init_devices(devices) -> None:
    """One process, several GPUs: one context per entry of `devices` (HIP device ids; an id may
    repeat).  See curdle_init_devices in include/curdle_msm.h."""
    arr = (C.c_int * len(devices))(*[int(d) for d in devices])
    _check(_init_devices(arr, len(devices)))


def device_count() -> int:
    return int(_device_count())


def set_device(ordinal: int) -> None:
    """The calling thread's current context (0 <= ordinal < device_count())."""
    _check(_set_device(int(ordinal)))


def get_device() -> int:
    return int(_get_device())


def get_device_selection() -> int:
    """The ordinal this thread selected with set_device, or -1 if it has made no selection."""
    return int(_get_device_selection())


def stat_spread_calls() -> int:
    return int(_stat_spread_calls())


# flags of the *_ex entry points (include/curdle_msm.h)
MSM_ANY_CURVE_POINT, MSM_BASES_UNCHANGED = 1, 2


SPLIT_AUTO, SPLIT_WINDOWS, SPLIT_POINTS = 0, 1, 2


def msm_g1_replicated(d_points, d_scalars, n: int, split: int = SPLIT_AUTO) -> np.ndarray:
    """One MSM over inputs resident on EVERY configured context (d_points[i] / d_scalars[i]: raw
    device pointers on context i's GPU), split by Pippenger windows or point ranges, one host
    thread per device, partials summed on the host."""
    k = device_count()
    assert len(d_points) == k and len(d_scalars) == k, "one pointer per configured device"
    pp = (C.c_void_p * k)(*[int(p) for p in d_points])
    ps = (C.c_void_p * k)(*[int(p) for p in d_scalars])
    out = np.zeros(18, dtype=np.uint64)
    _check(_msm_g1_replicated(pp, ps, n, int(split), _ptr(out)))
    return out


def device_available() -> bool:
    return bool(_device_available())


def msm_g1(points, scalars, flags: int = 0) -> np.ndarray:
    """(*G1Jac).MultiExp on host arrays: points uint64[n,12], scalars uint64[n,4] -> uint64[18].
    flags: MSM_ANY_CURVE_POINT = no endomorphism, gnark's result for bases outside the subgroup too."""
    points = _as_u64(points, 12)
    scalars = _as_u64(scalars, 4)
    n = points.shape[0] if points.size else 0
    ns = scalars.shape[0] if scalars.size else 0
    if n != ns:
        # gnark MultiExp: "len(points) != len(scalars)"
        raise CurdleError(EINVAL, "len(points) != len(scalars)")
    out = np.zeros(18, dtype=np.uint64)
    if flags:
        _check(_msm_g1_ex(_ptr(points), _ptr(scalars), n, int(flags), _ptr(out)))
    else:
        _check(_msm_g1(_ptr(points), _ptr(scalars), n, _ptr(out)))
    return out


def msm_g1_device(d_points: int, d_scalars: int, n: int, stream: int = 0, window_bits: int = 0,
                  win_begin: int = 0, win_end: int = -1, flags: int = 0) -> np.ndarray:
    """MSM on device-resident inputs (raw device pointers, e.g. torch_tensor.data_ptr()).
    flags: MSM_ANY_CURVE_POINT, MSM_BASES_UNCHANGED (the library keeps its converted copy of d_points)."""
    out = np.zeros(18, dtype=np.uint64)
    if flags:
        _check(_msm_g1_device_windows_ex(d_points, d_scalars, n, window_bits, win_begin, win_end, int(flags), _ptr(out),
                                         stream or None))
    else:
        _check(_msm_g1_device_windows(d_points, d_scalars, n, window_bits, win_begin, win_end, _ptr(out),
                                      stream or None))
    return out


def msm_g1_device_submit(d_points: int, d_scalars: int, n: int, window_bits: int = 0, win_begin: int = 0,
                         win_end: int = -1, flags: int = 0) -> int:
    """Enqueue one MSM (or window range) and return a ticket; see msm_wait()."""
    t = C.c_int(-1)
    if flags:
        _check(_msm_submit_ex(d_points, d_scalars, n, window_bits, win_begin, win_end, int(flags), C.byref(t)))
    else:
        _check(_msm_submit(d_points, d_scalars, n, window_bits, win_begin, win_end, C.byref(t)))
    return t.value


def msm_forget_bases(d_points: int) -> None:
    """Drops the converted copy a MSM_BASES_UNCHANGED call left for d_points."""
    _check(_msm_forget_bases(d_points))


class DBases:
    """A resident, pre-converted base set (curdle_dbases): the plain MSM over it uploads and converts
    nothing per call."""

    def __init__(self, points):
        pts = _as_u64(points, 12)
        self.n = len(pts)
        self._h = C.c_void_p()
        _check(_dbases_create(_ptr(pts), self.n, C.byref(self._h)))

    def free(self):
        if self._h:
            _dbases_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass

    def msm(self, d_scalars: int, n: int = None, window_bits: int = 0, win_begin: int = 0, win_end: int = -1) -> np.ndarray:
        """scalars in device memory; a window range gives the scaled partial."""
        out = np.zeros(18, dtype=np.uint64)
        _check(_msm_dbases_windows(self._h, d_scalars, self.n if n is None else n, window_bits, win_begin, win_end, _ptr(out)))
        return out

    def msm_host(self, scalars) -> np.ndarray:
        sc = _as_u64(scalars, 4)
        out = np.zeros(18, dtype=np.uint64)
        _check(_msm_dbases_host(self._h, _ptr(sc), len(sc), _ptr(out)))
        return out

    def submit(self, d_scalars: int, n: int = None, window_bits: int = 0, win_begin: int = 0, win_end: int = -1) -> int:
        t = C.c_int(-1)
        _check(_msm_dbases_submit(self._h, d_scalars, self.n if n is None else n, window_bits, win_begin, win_end, C.byref(t)))
        return t.value


def msm_wait(ticket: int) -> np.ndarray:
    out = np.zeros(18, dtype=np.uint64)
    _check(_msm_wait(ticket, _ptr(out)))
    return out


def window_bits(n: int) -> int:
    return _window_bits(n)


def num_windows(n: int, c: int = 0, flags: int = 0) -> int:
    """W of the plan a call with these flags runs: ceil(127 / c), or ceil(255 / c) with MSM_ANY_CURVE_POINT."""
    rc = _num_windows_ex(n, c, int(flags))
    if rc < 0:
        _check(rc)
    return rc


def window_widths(n: int, c: int = 0, flags: int = 0):
    """Widths (bits) of the Pippenger windows for (n, c, flags), lowest first; the top one is unsigned."""
    buf = (C.c_int * 64)()
    W = _window_widths_ex(n, c, int(flags), buf)
    if W < 0:
        _check(W)
    return [int(buf[i]) for i in range(W)]


def msm_g1_batch_device(d_points: int, d_scalars: int, offsets, stream: int = 0) -> np.ndarray:
    """k independent MSMs over device-resident, concatenated inputs -> uint64[k, 18]."""
    offs = np.ascontiguousarray(offsets, dtype=np.uint64)
    k = len(offs) - 1
    out = np.zeros((k, 18), dtype=np.uint64)
    _check(_msm_batch_device(d_points, d_scalars, _ptr(offs), k, _ptr(out), stream or None))
    return out


def g1_sum(jac_points) -> np.ndarray:
    jac_points = _as_u64(jac_points, 18)
    k = jac_points.shape[0] if jac_points.size else 0
    out = np.zeros(18, dtype=np.uint64)
    _check(_g1_sum(_ptr(jac_points), k, _ptr(out)))
    return out


def msm_g1_batch(points, scalars, offsets) -> np.ndarray:
    points = _as_u64(points, 12)
    scalars = _as_u64(scalars, 4)
    offs = np.ascontiguousarray(offsets, dtype=np.uint64)  # size_t
    k = len(offs) - 1
    out = np.zeros((k, 18), dtype=np.uint64)
    _check(_msm_batch(_ptr(points), _ptr(scalars), _ptr(offs), k, _ptr(out)))
    return out


def msm_g1_multi(points_sets, scalars) -> np.ndarray:
    sets = [_as_u64(p, 12) for p in points_sets]
    scalars = _as_u64(scalars, 4)
    n = scalars.shape[0] if scalars.size else 0
    for s in sets:
        if (s.shape[0] if s.size else 0) != n:
            raise CurdleError(EINVAL, "len(points) != len(scalars)")
    arr = (_vp * len(sets))(*[s.ctypes.data for s in sets])
    out = np.zeros((len(sets), 18), dtype=np.uint64)
    _check(_msm_multi(C.cast(arr, _vp), len(sets), _ptr(scalars), n, _ptr(out)))
    return out


class Rand:
    """common.Rand (common/rand.go) through the library's host mirror."""

    def __init__(self, seed: int):
        self._h = _rand_new(seed)
        if not self._h:
            raise MemoryError("curdle_rand_new")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _rand_free is not None:      # module globals may be gone at interpreter exit
            _rand_free(h)

    def get_fr(self) -> np.ndarray:
        out = np.zeros(4, dtype=np.uint64)
        _check(_rand_get_fr(self._h, _ptr(out)))
        return out

    def get_frs(self, n: int) -> np.ndarray:
        return np.array([self.get_fr() for _ in range(n)], dtype=np.uint64).reshape(n, 4)

    def get_g1_affine(self) -> np.ndarray:
        out = np.zeros(12, dtype=np.uint64)
        _check(_rand_get_g1(self._h, _ptr(out)))
        return out

    def get_g1_affines(self, n: int) -> np.ndarray:
        return np.array([self.get_g1_affine() for _ in range(n)], dtype=np.uint64).reshape(n, 12)

    def generate_permutation(self, n: int) -> np.ndarray:
        out = np.zeros(n, dtype=np.uint32)
        _check(_rand_perm(self._h, n, _ptr(out)))
        return out


class MsmAccumulator:
    """msmaccumulator.MsmAccumulator (msmaccumulator/msmaccumulator.go:11-64)."""

    def __init__(self):
        self._h = _acc_new()
        if not self._h:
            raise MemoryError("curdle_acc_new")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _acc_free is not None:
            _acc_free(h)

    def accumulate_check(self, C_jac, x, v, rand: Rand) -> None:
        C_jac = _as_u64(C_jac)
        x = _as_u64(x, 4)
        v = _as_u64(v, 12)
        nx = x.shape[0] if x.size else 0
        nv = v.shape[0] if v.size else 0
        _check(_acc_check(self._h, _ptr(C_jac), _ptr(x), nx, _ptr(v), nv, rand._h))

    def accumulate_check_deferred(self, c_scalars, c_points, x, v, rand: Rand) -> None:
        """AccumulateCheck with C = sum_j c_scalars[j] * c_points[j] folded into the map."""
        cs = _as_u64(c_scalars, 4)
        cp = _as_u64(c_points, 12)
        x = _as_u64(x, 4)
        v = _as_u64(v, 12)
        n = lambda a: a.shape[0] if a.size else 0
        _check(_acc_check_deferred(self._h, _ptr(cs), _ptr(cp), n(cs), _ptr(x), n(x), _ptr(v), n(v), rand._h))

    def verify(self) -> bool:
        ok = C.c_int(0)
        _check(_acc_verify(self._h, C.byref(ok)))
        return bool(ok.value)

    @property
    def A_c(self) -> np.ndarray:
        out = np.zeros(18, dtype=np.uint64)
        _check(_acc_get_A_c(self._h, _ptr(out)))
        return out

    def num_bases(self) -> int:
        return _acc_num_bases(self._h)

    def export(self):
        n = self.num_bases()
        pts = np.zeros((n, 12), dtype=np.uint64)
        sc = np.zeros((n, 4), dtype=np.uint64)
        if n:
            _check(_acc_export(self._h, _ptr(pts), _ptr(sc)))
        return pts, sc


def profile_enable(on=True) -> None:
    """True / 1: HIP events around every kernel; 2: around the dominant kernel only; False / 0: off."""
    _check(_profile_enable(int(on)))


def profile_last() -> dict:
    p = _Profile()
    _check(_profile_last(C.byref(p)))
    return {
        "kernels": {p.name[i].decode(): float(p.ms[i]) for i in range(p.n_kernels)},
        "window_bits": p.window_bits,
        "num_windows": p.num_windows,
        "entries": int(p.entries),
        "fragments": int(p.fragments),
    }


def selftest_shape(op: int):
    """(words read, words written) per item of selftest operation `op`: the library's own table."""
    iw, ow = C.c_uint32(0), C.c_uint32(0)
    _check(_selftest_shape(op, C.byref(iw), C.byref(ow)))
    return iw.value, ow.value


def selftest_op(op: int, inp: np.ndarray, on_device: bool) -> np.ndarray:
    """inp: uint32[n, in_width] -> uint32[n, out_width] (see curdle_selftest_op)."""
    iw, ow = selftest_shape(op)
    inp = np.ascontiguousarray(inp, dtype=np.uint32)
    assert inp.ndim == 2 and inp.shape[1] == iw, inp.shape
    out = np.zeros((inp.shape[0], ow), dtype=np.uint32)
    _check(_selftest_op(op, _ptr(inp), inp.shape[0], _ptr(out), 1 if on_device else 0))
    return out


def int_to_limbs(v: int, n: int = 4) -> np.ndarray:
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def synth_points_walk_device(k: int, q: int, n: int, d_out: int) -> None:
    """P_i = (k + i*q) * G for i < n, written as gnark G1Affine into device memory at d_out
    (n * 96 bytes).  k, q are canonical integers < r."""
    ka, qa = int_to_limbs(k), int_to_limbs(q)
    _check(_synth_walk(_ptr(ka), _ptr(qa), n, d_out))


# ---------------------------------------------------------------------------
# Protocol layers (host restatement of curdleproof.Prove / Verify; include/curdle_msm.h)
# ---------------------------------------------------------------------------
class CRS:
    """curdleproof.CRS from GenerateCRS(ell, rand) (crs.go:20)."""

    def __init__(self, ell: int, rand: Rand):
        self._h = _crs_generate(ell, rand._h)
        if not self._h:
            raise MemoryError("curdle_crs_generate")
        self.ell = ell

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _crs_free is not None:
            _crs_free(h)


def shuffle_permute_commit(crs: CRS, Rs, Ss, perm, k, rand: Rand):
    """common.ShufflePermuteCommit (common/util.go:45) -> (Ts, Us, M, rs_m)."""
    Rs, Ss = _as_u64(Rs, 12), _as_u64(Ss, 12)
    perm = np.ascontiguousarray(perm, dtype=np.uint32)
    k = _as_u64(k)
    ell = crs.ell
    Ts = np.zeros((ell, 12), dtype=np.uint64)
    Us = np.zeros((ell, 12), dtype=np.uint64)
    M = np.zeros(18, dtype=np.uint64)
    rs_m = np.zeros((4, 4), dtype=np.uint64)
    _check(_spc(crs._h, _ptr(Rs), _ptr(Ss), ell, _ptr(perm), _ptr(k), rand._h, _ptr(Ts), _ptr(Us), _ptr(M), _ptr(rs_m)))
    return Ts, Us, M, rs_m


def prove(crs: CRS, Rs, Ss, Ts, Us, M, perm, k, rs_m, rand: Rand) -> bytes:
    """curdleproof.Prove (curdleproof.go:38); returns the serialised proof."""
    Rs, Ss, Ts, Us = (_as_u64(a, 12) for a in (Rs, Ss, Ts, Us))
    M, k, rs_m = _as_u64(M), _as_u64(k), _as_u64(rs_m)
    perm = np.ascontiguousarray(perm, dtype=np.uint32)
    cap = 1 << 16
    buf = np.zeros(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    _check(_prove(crs._h, _ptr(Rs), _ptr(Ss), _ptr(Ts), _ptr(Us), crs.ell, _ptr(M), _ptr(perm), _ptr(k), _ptr(rs_m),
                  rand._h, _ptr(buf), cap, C.byref(n)))
    return bytes(buf[: n.value])


def verify(crs: CRS, proof: bytes, Rs, Ss, Ts, Us, M, rand: Rand) -> bool:
    """curdleproof.Verify (curdleproof.go:199): the accept bit; raises CurdleError for (false, err)."""
    Rs, Ss, Ts, Us = (_as_u64(a, 12) for a in (Rs, Ss, Ts, Us))
    M = _as_u64(M)
    pb = np.frombuffer(proof, dtype=np.uint8).copy()
    ok = C.c_int(0)
    _check(_verify(crs._h, _ptr(pb), len(pb), _ptr(Rs), _ptr(Ss), _ptr(Ts), _ptr(Us), crs.ell, _ptr(M), rand._h,
                   C.byref(ok)))
    return bool(ok.value)


class Proof:
    """A decoded curdleproof.Proof (Proof.FromReader, curdleproof.go:320: every point curve- and
    subgroup-checked).  verify_proof() is the reference's Verify(proof Proof, ...) on it."""

    def __init__(self, data: bytes):
        pb = np.frombuffer(data, dtype=np.uint8).copy()
        h = _vp()
        _check(_proof_from_bytes(_ptr(pb), len(pb), C.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _proof_free is not None:
            _proof_free(h)


def verify_proof(crs: CRS, proof: Proof, Rs, Ss, Ts, Us, M, rand: Rand) -> bool:
    Rs, Ss, Ts, Us = (_as_u64(a, 12) for a in (Rs, Ss, Ts, Us))
    M = _as_u64(M)
    ok = C.c_int(0)
    _check(_verify_proof(crs._h, proof._h, _ptr(Rs), _ptr(Ss), _ptr(Ts), _ptr(Us), crs.ell, _ptr(M), rand._h, C.byref(ok)))
    return bool(ok.value)


class PreparedVerify:
    """The arguments of curdle_verify_proof marshalled once (the instance arrays as C pointers), so that run() is
    the C call and nothing else: what a benchmark of the verifier should time -- the reference's BenchmarkVerifier
    (curdleproof_test.go:210-237) times Verify on values that are already in memory, and five numpy -> ctypes
    conversions per call are ~15 us of a 0.8 ms verification."""

    def __init__(self, crs: CRS, proof: Proof, Rs, Ss, Ts, Us, M):
        self._keep = [_as_u64(a, 12) for a in (Rs, Ss, Ts, Us)] + [_as_u64(M)]
        self._args = (crs._h, proof._h, *[_ptr(a) for a in self._keep[:4]], crs.ell, _ptr(self._keep[4]))
        self._crs, self._proof = crs, proof      # keep the handles alive
        self._ok = C.c_int(0)
        self._okref = C.byref(self._ok)

    def run(self, rand: Rand) -> bool:
        _check(_verify_proof(*self._args, rand._h, self._okref))
        return bool(self._ok.value)


class PreparedVerifyBatch:
    """The arguments of curdle_verify_batch marshalled once (pointer tables over contiguous
    arrays), so that run() is the C call and nothing else -- what a benchmark should time, and
    what a caller verifying the same batch layout repeatedly wants."""

    def __init__(self, proofs, Rs, Ss, Ts, Us, Ms):
        k = self.k = len(proofs)
        self._keep = []

        def ptr_array(arrs, width):
            a = [_as_u64(x, width) for x in arrs]
            self._keep.append(a)
            return (C.c_void_p * k)(*[x.ctypes.data for x in a])

        self._pbufs = [np.frombuffer(p, dtype=np.uint8).copy() for p in proofs]
        self._pp = (C.c_void_p * k)(*[b.ctypes.data for b in self._pbufs])
        self._lens = (C.c_size_t * k)(*[len(b) for b in self._pbufs])
        self._ms = np.ascontiguousarray(np.stack([_as_u64(m) for m in Ms]) if k else np.zeros((0, 18), dtype=np.uint64))
        self._inst = [ptr_array(v, 12) for v in (Rs, Ss, Ts, Us)]

    def run(self, crs: CRS, rand: Rand, nthreads: int = 8):
        oks = (C.c_int * self.k)()
        _check(_verify_batch(crs._h, self.k, self._pp, self._lens, *self._inst, crs.ell, _ptr(self._ms), rand._h,
                             int(nthreads), oks))
        return [bool(v) for v in oks]


def verify_batch(crs: CRS, proofs, Rs, Ss, Ts, Us, Ms, rand: Rand, nthreads: int = 8):
    """Cross-proof batch verification: k proofs over one CRS, one shared accumulator, one MSM.
    proofs: list of bytes; Rs/Ss/Ts/Us: lists of (ell, 12) arrays; Ms: list of 18-limb points.
    Returns the list of accept bits (exact: a failing batch is settled proof by proof)."""
    return PreparedVerifyBatch(proofs, Rs, Ss, Ts, Us, Ms).run(crs, rand, nthreads)


# ---- whisk package (whisk/whisk.go, whisk/types.go): trackers are 96-byte strings rG || krG ----
WHISK_ELL = 124
WHISK_TRACKER_PROOF_SIZE = 128
WHISK_SHUFFLE_PROOF_SIZE = 4576


def _bytes_arr(b: bytes) -> np.ndarray:
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


def whisk_is_valid_shuffle_proof(crs: CRS, pre_trackers, post_trackers, proof: bytes, rand: Rand) -> bool:
    """IsValidWhiskShuffleProof (whisk.go:20): trackers are lists of 96-byte strings."""
    if len(proof) != WHISK_SHUFFLE_PROOF_SIZE:
        raise ValueError("a whisk shuffle proof is %d bytes" % WHISK_SHUFFLE_PROOF_SIZE)
    pre, post, pb = _bytes_arr(b"".join(pre_trackers)), _bytes_arr(b"".join(post_trackers)), _bytes_arr(proof)
    ok = C.c_int(0)
    _check(_whisk_valid_shuffle(crs._h, _ptr(pre), _ptr(post), len(pre_trackers), len(post_trackers), _ptr(pb), rand._h,
                                C.byref(ok)))
    return bool(ok.value)


class PreparedWhiskBatch:
    """The arguments of curdle_whisk_is_valid_shuffle_proof_batch marshalled once; run() is the
    C call and nothing else (see PreparedVerifyBatch)."""

    def __init__(self, pre_sets, post_sets, proofs):
        k = self.k = len(proofs)
        n = self.n = len(pre_sets[0]) if k else 0
        self._pre = [_bytes_arr(b"".join(s)) for s in pre_sets]
        self._post = [_bytes_arr(b"".join(s)) for s in post_sets]
        self._pb = [_bytes_arr(p) for p in proofs]
        if any(len(a) != 96 * n for a in self._pre + self._post) or any(len(p) != WHISK_SHUFFLE_PROOF_SIZE for p in self._pb):
            raise ValueError("every tracker set needs the same length; proofs are %d bytes" % WHISK_SHUFFLE_PROOF_SIZE)
        arr = lambda bufs: (C.c_void_p * k)(*[b.ctypes.data for b in bufs])
        self._ptrs = (arr(self._pre), arr(self._post), arr(self._pb))

    def run(self, crs: CRS, rand: Rand, nthreads: int = 16):
        oks = (C.c_int * self.k)()
        _check(_whisk_valid_shuffle_batch(crs._h, self.k, self._ptrs[0], self._ptrs[1], self.n, self._ptrs[2], rand._h,
                                          int(nthreads), oks))
        return [bool(v) for v in oks]


def whisk_is_valid_shuffle_proof_batch(crs: CRS, pre_sets, post_sets, proofs, rand: Rand, nthreads: int = 16):
    """k shuffle proofs at once: pre_sets / post_sets are lists of tracker lists (all the same
    length), proofs a list of 4,576-byte strings.  Returns the list of accept bits."""
    return PreparedWhiskBatch(pre_sets, post_sets, proofs).run(crs, rand, nthreads)


def whisk_generate_shuffle_proof(crs: CRS, pre_trackers, rand: Rand):
    """GenerateWhiskShuffleProof (whisk.go:63) -> (post_trackers, proof bytes)."""
    pre = _bytes_arr(b"".join(pre_trackers))
    post = np.zeros(96 * len(pre_trackers), dtype=np.uint8)
    proof = np.zeros(WHISK_SHUFFLE_PROOF_SIZE, dtype=np.uint8)
    _check(_whisk_gen_shuffle(crs._h, _ptr(pre), len(pre_trackers), rand._h, _ptr(post), _ptr(proof)))
    pb = post.tobytes()
    return [pb[96 * i: 96 * (i + 1)] for i in range(len(pre_trackers))], proof.tobytes()


def whisk_is_valid_tracker_proof(tracker: bytes, k_commitment: bytes, proof: bytes) -> bool:
    """IsValidWhiskTrackerProof (whisk.go:116)."""
    if len(tracker) != 96 or len(k_commitment) != 48 or len(proof) != WHISK_TRACKER_PROOF_SIZE:
        raise ValueError("tracker 96 B, k commitment 48 B, tracker proof 128 B")
    t, kc, pb = _bytes_arr(tracker), _bytes_arr(k_commitment), _bytes_arr(proof)
    ok = C.c_int(0)
    _check(_whisk_valid_tracker(_ptr(t), _ptr(kc), _ptr(pb), C.byref(ok)))
    return bool(ok.value)


def whisk_generate_tracker_proof(tracker: bytes, k, rand: Rand) -> bytes:
    """GenerateWhiskTrackerProof (whisk.go:149); k = Montgomery fr limbs."""
    t, kk = _bytes_arr(tracker), _as_u64(k)
    out = np.zeros(WHISK_TRACKER_PROOF_SIZE, dtype=np.uint8)
    _check(_whisk_gen_tracker(_ptr(t), _ptr(kk), rand._h, _ptr(out)))
    return out.tobytes()


_set_dev_acc = _sig("curdle_verify_set_device_acc", C.c_int, C.c_int)
_export_acc = _sig("curdle_verify_export_accumulator", C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp, C.c_int,
                   _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int))


def verify_set_device_acc(on: bool) -> bool:
    """Keep curdleproof.Verify's accumulator on the device (default) or on the host mirror;
    returns the previous setting."""
    return bool(_set_dev_acc(1 if on else 0))


def verify_export_accumulator(crs: "CRS", proof: "Proof", Rs, Ss, Ts, Us, M, rand: "Rand", device: bool):
    """(points[n, 12], scalars[n, 4], accept) accumulated by one verification before its final
    MSM, from the host mirror (device=False) or the device accumulator (device=True)."""
    Rs, Ss, Ts, Us = (_as_u64(a, 12) for a in (Rs, Ss, Ts, Us))
    M = _as_u64(M)
    cap = 6 * crs.ell + 8192
    pts = np.zeros((cap, 12), dtype=np.uint64)
    sc = np.zeros((cap, 4), dtype=np.uint64)
    n, ok = C.c_size_t(0), C.c_int(0)
    _check(_export_acc(crs._h, proof._h, _ptr(Rs), _ptr(Ss), _ptr(Ts), _ptr(Us), crs.ell, _ptr(M), rand._h,
                       1 if device else 0, _ptr(pts), _ptr(sc), cap, C.byref(n), C.byref(ok)))
    return pts[: n.value].copy(), sc[: n.value].copy(), bool(ok.value)


def verify_set_eager(eager: bool) -> bool:
    """Evaluate the verifier's check points eagerly (the reference's order of operations)
    instead of deferring them into the accumulator's one MSM; returns the previous mode."""
    return bool(_verify_set_eager(1 if eager else 0))


def proof_reencode(proof: bytes) -> bytes:
    pb = np.frombuffer(proof, dtype=np.uint8).copy()
    out = np.zeros(len(pb) + 64, dtype=np.uint8)
    n = C.c_size_t(0)
    _check(_reencode(_ptr(pb), len(pb), _ptr(out), len(out), C.byref(n)))
    return bytes(out[: n.value])


_fr_ip = _sig("curdle_fr_inner_product", C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp)


def fr_inner_product(a, b) -> np.ndarray:
    """common.IPA (reference common/util.go:26): Montgomery-form Fr vectors in, Montgomery Fr out."""
    a = _as_u64(a, 4)
    b = _as_u64(b, 4)
    out = np.zeros(4, dtype=np.uint64)
    _check(_fr_ip(_ptr(a), len(a), _ptr(b), len(b), _ptr(out)))
    return out


def merlin_test_vector(protocol: bytes, label: bytes, msg: bytes, challenge_label: bytes, n: int) -> bytes:
    m = np.frombuffer(msg, dtype=np.uint8).copy()
    out = np.zeros(n, dtype=np.uint8)
    _check(_merlin_tv(protocol, label, _ptr(m), len(m), challenge_label, _ptr(out), n))
    return bytes(out)


_decompress_batch = _sig("curdle_g1_decompress_batch", C.c_int, _vp, C.c_size_t, C.c_int, _vp, _vp)
DECODE_OK, DECODE_INFINITY, DECODE_BAD_ENCODING, DECODE_NOT_ON_CURVE, DECODE_NOT_IN_SUBGROUP = range(5)


_decompress_begin = _sig("curdle_g1_decompress_begin", C.c_int, _vp, C.c_size_t, _vp, _vp, C.POINTER(C.c_int))
_decompress_finish = _sig("curdle_g1_decompress_finish", C.c_int, C.c_int, _vp)


def g1_decompress_begin(data: bytes):
    """Two-step decode: returns (points, preliminary status, ticket); the subgroup test keeps
    running on the GPU until g1_decompress_finish(ticket, n) returns the final status bytes."""
    b = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    n = len(b) // 48
    out = np.zeros((n, 12), dtype=np.uint64)
    st = np.zeros(n, dtype=np.uint8)
    t = C.c_int(-1)
    _check(_decompress_begin(_ptr(b), n, _ptr(out), _ptr(st), C.byref(t)))
    return out, st, t.value


def g1_decompress_finish(ticket: int, n: int) -> np.ndarray:
    st = np.zeros(n, dtype=np.uint8)
    _check(_decompress_finish(ticket, _ptr(st)))
    return st


def g1_decompress_batch(data: bytes, subgroup_check: bool = True):
    """n x 48 bytes of compressed points -> ((n, 12) gnark affine points, (n,) status bytes), on the GPU."""
    b = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    if len(b) % 48:
        raise ValueError("compressed G1 points are 48 bytes each")
    n = len(b) // 48
    out = np.zeros((n, 12), dtype=np.uint64)
    st = np.zeros(n, dtype=np.uint8)
    _check(_decompress_batch(_ptr(b), n, 1 if subgroup_check else 0, _ptr(out), _ptr(st)))
    return out, st


_scalar_mul_batch = _sig("curdle_g1_scalar_mul_batch", C.c_int, _vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp)


def g1_scalar_mul_batch(points, scalars, addends=None) -> np.ndarray:
    """out[i] = addends[i] + scalars[i] * points[i] on the GPU; scalars (n, 4) or one shared (4,)."""
    points = _as_u64(points, 12)
    n = points.shape[0] if points.size else 0
    sc = _as_u64(scalars, 4)
    ns = (1 if sc.ndim == 1 else sc.shape[0]) if sc.size else 0
    add = _as_u64(addends, 12) if addends is not None else None
    out = np.zeros((n, 12), dtype=np.uint64)
    _check(_scalar_mul_batch(_ptr(points), _ptr(sc), ns, _ptr(add) if add is not None else None, n, _ptr(out)))
    return out


def g1_compress(jac) -> bytes:
    jac = _as_u64(jac)
    out = np.zeros(48, dtype=np.uint8)
    _check(_g1_compress(_ptr(jac), _ptr(out)))
    return bytes(out)


def g1_decompress(data: bytes, subgroup_check: bool = True) -> np.ndarray:
    b = np.frombuffer(data, dtype=np.uint8).copy()
    out = np.zeros(18, dtype=np.uint64)
    _check(_g1_decompress(_ptr(b), 1 if subgroup_check else 0, _ptr(out)))
    return out
