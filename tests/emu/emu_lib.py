"""TEST INFRASTRUCTURE ONLY: loads the host build of the kernels (tests/emu/build_emu.py) with the C-ABI prototypes of
kornia_amd/_native.py, for calls on host buffers.  Nothing under kornia_amd/ imports this."""
from __future__ import annotations

import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)

_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        import build_emu

        from kornia_amd import _native

        h = ctypes.CDLL(build_emu.build())
        h.km_abi_version.restype = ctypes.c_int
        h.km_last_error.restype = ctypes.c_char_p
        for name, argtypes in _native._PROTOTYPES.items():
            fn = getattr(h, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        h.emu_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
        h.emu_set_schedule.argtypes = [ctypes.c_int, ctypes.c_ulonglong]
        h.emu_set_schedule.restype = None
        h.emu_set_glds.argtypes = [ctypes.c_int]
        h.emu_set_glds.restype = None
        _lib = h
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what}: rc={rc}: {lib().km_last_error().decode()}")


def stats() -> dict:
    out = (ctypes.c_ulonglong * 4)()
    lib().emu_stats(out)
    return {"launches": out[0], "workgroups": out[1], "dead_lane_reads": out[2], "misaligned_vector_accesses": out[3]}


SCHEDULES = {"forward": 0, "reverse": 1, "random": 2, "lanes": 3}


def set_schedule(mode: str, seed: int = 1) -> None:
    """Order in which ready work-items resume: 'forward', 'reverse' (waves), 'random' (waves shuffled), 'lanes' (everything shuffled)."""
    lib().emu_set_schedule(SCHEDULES[mode], seed)


def set_glds(deferred: bool) -> None:
    """LDS-DMA model: False = the piece lands at the request (earliest legal), True = at the requesting lane's KM_VMCNT0(), shuffled (latest legal)."""
    lib().emu_set_glds(1 if deferred else 0)
