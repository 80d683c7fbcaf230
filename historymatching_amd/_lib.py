"""ctypes binding of libhm_amd.so (the C ABI declared in include/hm_abi.h).

There is no CPU fallback: if the shared library is missing, or no HIP device is visible when a
context is requested, the call raises.  Nothing here imports ``oracle/``.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("HM_AMD_LIB", _HERE / "libhm_amd.so"))


class HmError(RuntimeError):
    """Raised when a C-ABI call returns nonzero (message = hm_last_error())."""


class hm_stats(C.Structure):
    _fields_ = [
        ("ms_total", C.c_double),
        ("ms_pressure", C.c_double),
        ("ms_saturation", C.c_double),
        ("ms_update", C.c_double),
        ("mean_nts", C.c_double),
        ("n_pressure_launches", C.c_longlong),
        ("n_saturation_launches", C.c_longlong),
        ("member_steps", C.c_longlong),
        ("mean_n_cg", C.c_double),
        ("ms_comm", C.c_double),
    ]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_vp, _ip, _dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/hm_abi.h declares
SIGNATURES = {
    "hm_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "hm_destroy": (None, [_vp]),
    "hm_device_count": (C.c_int, []),
    "hm_last_error": (C.c_char_p, []),
    "hm_device_name": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "hm_abi_version": (C.c_int, []),
    "hm_copy_to_host": (C.c_int, [_vp, _vp, _vp, C.c_longlong]),
    "hm_copy_to_device": (C.c_int, [_vp, _vp, _vp, C.c_longlong]),
    "hm_comm_unique_id": (C.c_int, [C.c_char_p]),
    "hm_comm_probe": (C.c_int, []),
    "hm_upd_chain_fallbacks": (C.c_int, [_vp]),
    "hm_comm_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p, C.POINTER(_vp)]),
    "hm_comm_destroy": (None, [_vp]),
    "hm_comm_rank": (C.c_int, [_vp]),
    "hm_comm_world_size": (C.c_int, [_vp]),
    "hm_comm_all_reduce": (C.c_int, [_vp, _vp, C.c_longlong, C.c_int, C.c_int]),
    "hm_comm_all_gather": (C.c_int, [_vp, _vp, C.c_longlong, C.c_int]),
    "hm_comm_broadcast": (C.c_int, [_vp, _vp, C.c_longlong, C.c_int, C.c_int]),
    "hm_comm_group_start": (C.c_int, [_vp]),
    "hm_comm_group_end": (C.c_int, [_vp]),
    "hm_comm_sync": (C.c_int, [_vp]),
    "hm_forward_batched": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _vp, C.c_int, _vp,
                                     C.c_int, _ip, _dp, C.c_int, C.c_int, _ip, _dp, C.c_int, C.c_double, C.c_int,
                                     C.c_double, C.c_double, C.c_double, C.c_double, _dp, C.c_int, C.c_int, _vp, _vp,
                                     _ip, C.POINTER(hm_stats)]),
    "hm_fwd_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _ip, _dp, C.c_int,
                                C.c_int, _ip, _dp, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double,
                                C.c_double, _dp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "hm_fwd_destroy": (None, [_vp]),
    "hm_fwd_set_inputs": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "hm_fwd_set_perm_y": (C.c_int, [_vp, _vp, C.c_int]),
    "hm_fwd_run": (C.c_int, [_vp, C.c_int, C.c_int]),
    "hm_fwd_sync": (C.c_int, [_vp, C.POINTER(hm_stats)]),
    "hm_fwd_nd_fallbacks": (C.c_longlong, [_vp]),
    "hm_fwd_team_retries": (C.c_longlong, [_vp]),
    "hm_fwd_slab_redos": (C.c_longlong, [_vp]),
    "hm_fwd_set_debug": (C.c_int, [_vp, C.c_char_p, C.c_longlong]),
    "hm_fwd_get_outputs": (C.c_int, [_vp, _vp, _vp, _ip]),
    "hm_fwd_run_to_host": (C.c_int, [_vp, _vp, _vp, _ip, C.POINTER(hm_stats)]),
    "hm_fwd_set_variant": (C.c_int, [_vp, C.c_int, C.c_int]),
    "hm_fwd_set_solver": (C.c_int, [_vp, C.c_double, C.c_int]),
    "hm_fwd_set_member_wells": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "hm_fwd_set_inputs_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "hm_fwd_pressure_only": (C.c_int, [_vp, C.c_int]),
    "hm_fwd_saturation_only": (C.c_int, [_vp, C.c_int]),
    "hm_fwd_get_field": (C.c_int, [_vp, C.c_char_p, _vp]),
    "hm_fwd_set_field": (C.c_int, [_vp, C.c_char_p, _vp]),
    "hm_fwd_device_ptr": (_vp, [_vp, C.c_char_p]),
    "hm_debug_mfma_f64": (C.c_int, [_vp, _dp, _dp, _dp]),
    "hm_debug_fracflow32_check": (C.c_int, [_vp, C.POINTER(C.c_ulonglong)]),
    "hm_debug_fracflow64_check": (C.c_int, [_vp, C.POINTER(C.c_ulonglong)]),
    "hm_debug_nd_tables": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_longlong), _ip, _ip, C.POINTER(C.c_short), C.POINTER(C.c_short)]),
    "hm_es_update": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp,
                               C.POINTER(hm_stats)]),
    "hm_center": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]),
    "hm_es_update_loc": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double,
                                   C.c_int, _vp, C.POINTER(hm_stats)]),
    "hm_upd_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "hm_upd_destroy": (None, [_vp]),
    "hm_upd_set_inputs": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double]),
    "hm_upd_phase": (C.c_int, [_vp, C.c_int]),
    "hm_upd_run": (C.c_int, [_vp]),
    "hm_upd_set_column_shard": (C.c_int, [_vp, C.c_int, C.c_int]),
    "hm_upd_all_reduce": (C.c_int, [_vp, _vp, C.c_int]),
    "hm_upd_run_comm": (C.c_int, [_vp, _vp]),
    "hm_recompose": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _vp]),
    "hm_sample_kron": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "hm_upd_set_inputs_device": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int]),
    "hm_upd_swap": (C.c_int, [_vp]),
    "hm_debug_spd_inverse": (C.c_int, [_vp, C.c_int, _dp, C.c_double, _dp]),
    "hm_debug_ldl_gain": (C.c_int, [_vp, C.c_int, C.c_int, _dp, C.c_double, _dp, C.POINTER(C.c_float)]),
    "hm_upd_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "hm_upd_reduce_buffer": (_vp, [_vp, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "hm_iles_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _dp, C.c_double, _dp, C.POINTER(_vp)]),
    "hm_iles_destroy": (None, [_vp]),
    "hm_iles_compose": (C.c_int, [_vp, _vp]),
    "hm_iles_step": (C.c_int, [_vp, _dp, _dp, C.c_double]),
    "hm_iles_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "hm_iles_get_weights": (C.c_int, [_vp, C.c_int, _dp]),
    "hm_iles_device_ptr": (_vp, [_vp, C.c_char_p]),
    "hm_upd_sync": (C.c_int, [_vp, C.POINTER(hm_stats)]),
    "hm_upd_get_output": (C.c_int, [_vp, _vp]),
    "hm_upd_device_ptr": (_vp, [_vp, C.c_char_p]),
}

_lib = None


def load():
    """Load libhm_amd.so and bind every declared symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise HmError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C historymatching_amd/csrc`). There is no CPU fallback."
        )
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().hm_last_error()
        raise HmError(f"{what}: {msg.decode() if msg else 'error %d' % rc}")


def ptr(a):
    """void* of a NumPy array (or None)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def as_c(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


class Context:
    """One HIP context (device + stream).  Reused process-wide per device."""

    _cache = {}

    def __init__(self, device=0):
        lib = load()
        h = C.c_void_p()
        check(lib.hm_create(int(device), C.byref(h)), "hm_create")
        self.handle = h
        self.device = int(device)
        self.lib = lib

    @classmethod
    def get(cls, device=None):
        if device is None:
            device = int(os.environ.get("HM_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        if device not in cls._cache:
            cls._cache[device] = cls(device)
        return cls._cache[device]

    _secondary = {}

    @classmethod
    def secondary(cls, device=None, index=0):
        """A further context (streams of its own) on `device`, one per (device, index) for the life of the process: the second
        member block of a large forward run uses it, and every forward-model closure shares it instead of making its own."""
        device = cls.get(device).device
        if (device, index) not in cls._secondary:
            cls._secondary[(device, index)] = cls(device)
        return cls._secondary[(device, index)]

    def device_count(self):
        return int(self.lib.hm_device_count())

    def close(self):
        """Release the streams, events and pinned staging buffers of a context made with ``Context(device)`` (the per-device
        contexts of ``Context.get`` live as long as the process)."""
        if self.handle and Context._cache.get(self.device) is not self and self not in Context._secondary.values():
            self.lib.hm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def name(self):
        buf = C.create_string_buffer(256)
        check(self.lib.hm_device_name(self.handle, buf, 256), "hm_device_name")
        return buf.value.decode()
