"""ctypes binding of the CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
It reuses the product's struct definitions and call sequence (strolle_amd.api.EngineBase)
with the `or_` prefix, so the same scene-building code drives oracle and product.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from strolle_amd.api import EngineBase, _Binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


_lib = None


def oracle_lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            build_oracle()
        _lib = C.CDLL(ORACLE_LIB)
    return _lib


def encode_output(frame: np.ndarray, fmt: int) -> np.ndarray:
    """or_encode_output: the composed RGBA32F frame as a render target of format `fmt` (StOutputFormat) would hold it."""
    frame = np.ascontiguousarray(frame, np.float32)
    pixels = frame.size // 4
    out = np.zeros(frame.shape, np.uint16 if fmt == 1 else np.uint8)
    fn = oracle_lib().or_encode_output
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    assert fn(frame.ctypes.data, pixels, int(fmt), out.ctypes.data) == 0
    return out


class OracleEngine(EngineBase):
    def __init__(self):
        super().__init__(_Binding(oracle_lib(), "or_", False))
        self._sizes = {}

    def create_camera(self, camera):
        h = super().create_camera(camera)
        self._sizes[h] = camera.size
        return h

    def update_camera(self, handle, camera):
        super().update_camera(handle, camera)
        self._sizes[handle] = camera.size

    def tick(self):
        self._check(self._b.tick(self._h))

    def set_pass_mask(self, mask: int):
        """or_debug_set_pass_mask: which reference passes render_camera executes (bit numbers = StPassBit)."""
        fn = self._b.lib.or_debug_set_pass_mask
        fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_uint64]
        self._check(fn(self._h, mask & 0xFFFFFFFFFFFFFFFF))

    def write_buffer(self, camera: int, buffer, data: np.ndarray):
        fn = self._b.lib.or_camera_write_buffer
        fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t]
        data = np.ascontiguousarray(data)
        self._check(fn(self._h, camera, int(buffer), data.ctypes.data, data.nbytes))

    def render_camera(self, handle, compose: bool = True):
        w, h = self._sizes[handle]
        out = np.zeros((h, w, 4), np.float32) if compose else None
        self._check(self._b.render_camera(self._h, handle, out.ctypes.data if compose else None))
        return out


def set_stack_limit(n: int = 24):
    """or_debug_set_stack_limit: pending entries a ray of the oracle may keep (24 = the reference's lib.rs:76; 64 stands for unbounded). Process-wide."""
    fn = oracle_lib().or_debug_set_stack_limit
    fn.restype = C.c_int; fn.argtypes = [C.c_int]
    assert fn(n) == 0, "stack limit out of range"


def stack_stats(reset: bool = False):
    """or_debug_stack_stats: (pushes dropped at the limit, deepest stack any ray reached) since the last reset."""
    fn = oracle_lib().or_debug_stack_stats
    fn.restype = C.c_int; fn.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_int), C.c_int]
    d, s = C.c_ulonglong(), C.c_int()
    fn(C.byref(d), C.byref(s), 1 if reset else 0)
    return int(d.value), int(s.value)
