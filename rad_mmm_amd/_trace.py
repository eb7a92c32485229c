"""rocprofv3 range markers (SURVEY §5 "tracing": the reference has none; a build has to add its own).

RADMMM_ROCTX=1 makes the host side emit roctx ranges -- `decoder.fwd`, `flow<i>.fwd`, `flow<i>.bwd`, `context_lstm.fwd/bwd`,
`loss` -- through librocprofiler-sdk-roctx (rocprofv3 --marker-trace records them; without a profiler attached the calls
are no-ops inside the library).  tools/kernel_stats.py --by-range groups a kernel trace by the range whose launches
produced each dispatch, which answers "which flow step / which phase" without ad-hoc probes.  Off (the default) every entry
point below is one attribute test: no library is loaded, nothing is called."""
from __future__ import annotations

import contextlib
import ctypes
import os

_lib = None
ENABLED = os.environ.get("RADMMM_ROCTX", "0") not in ("", "0")


def _load():
    global _lib
    if _lib is None:
        for name in ("librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                _lib = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _lib is None:
            raise RuntimeError("RADMMM_ROCTX=1 but no roctx library (librocprofiler-sdk-roctx.so / libroctx64.so) could be loaded")
        _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
        _lib.roctxRangePushA.restype = ctypes.c_int
        _lib.roctxRangePop.restype = ctypes.c_int
    return _lib


def push(name: str) -> None:
    if ENABLED:
        _load().roctxRangePushA(name.encode())


def pop() -> None:
    if ENABLED:
        _load().roctxRangePop()


@contextlib.contextmanager
def trace_range(name: str):
    """with trace_range("flow3.fwd"): ...  (host-side range around the launches; nests)"""
    if not ENABLED:
        yield
        return
    push(name)
    try:
        yield
    finally:
        pop()


def traced(name: str):
    """decorator: the call inside a range `name` (one attribute test when markers are off)"""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*a, **k):
            if not ENABLED:
                return fn(*a, **k)
            push(name)
            try:
                return fn(*a, **k)
            finally:
                pop()
        return wrapped
    return deco
