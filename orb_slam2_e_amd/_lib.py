"""Build + load liborbslam_hip.so (the C-ABI of include/*.h) through ctypes.

There is no CPU fallback: if the library is missing it is built with hipcc, and
if it cannot be loaded the import fails loudly.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "liborbslam_hip.so")
_LIB = None


class OrbxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"orbslam_hip error {code}: {msg}")
        self.code = code


def build(force=False):
    csrc = os.path.join(_HERE, "csrc")
    args = ["make", "-C", csrc] + (["-B"] if force else [])
    # one builder at a time (several ranks of one run may find the library missing together)
    import fcntl
    with open(os.path.join(_HERE, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            subprocess.check_call(args, stdout=subprocess.DEVNULL)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    if not os.path.exists(SO_PATH):
        raise RuntimeError("liborbslam_hip.so was not produced by the build")
    return SO_PATH


def _one_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  The loader
    de-duplicates by SONAME only if torch's copy is loaded first, so import torch
    before dlopen()ing our library: one HIP runtime per process, and torch streams /
    device pointers are then valid inside the C-ABI calls."""
    if os.environ.get("ORBX_NO_TORCH_PRELOAD"):
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def lib():
    global _LIB
    if _LIB is None:
        # ORBX_LIB: another build of the SAME sources (tools/asan_host.sh: the host-only ASan / UBSan build)
        path = os.environ.get("ORBX_LIB") or SO_PATH
        if path == SO_PATH and not os.path.exists(SO_PATH):
            build()
        _one_hip_runtime()
        _LIB = C.CDLL(path)
        _LIB.orbx_last_error.restype = C.c_char_p
    return _LIB


def check(rc):
    if rc != 0:
        raise OrbxError(rc, lib().orbx_last_error().decode())
    return rc


_from_buffer, _addressof, _void_p = C.c_char.from_buffer, C.addressof, C.c_void_p


def ptr(a):
    """Address of a numpy array's data as c_void_p.  `a.ctypes.data_as(c_void_p)` costs ~3 us a piece -- with seven arrays a call
    that is more than a 50-us search; the buffer protocol gives the same address in ~0.4 us (the array is kept alive by the returned object, as data_as does).  Read-only and empty arrays (which
    the writable-buffer request refuses) take the slow way."""
    try:
        v = _void_p(_addressof(_from_buffer(a)))
    except (TypeError, ValueError, BufferError):
        return a.ctypes.data_as(_void_p)
    v._array = a        # as data_as does: the pointer keeps its array alive (callers pass temporaries)
    return v


def bind(fn, argtypes):
    """Sets a C function's argtypes once (assigning them costs ~1.5 us, so not per call)."""
    if fn.argtypes is None:
        fn.argtypes = argtypes
    return fn
