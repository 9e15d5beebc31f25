"""ctypes bindings for the two checkers.  TEST INFRASTRUCTURE ONLY.

* ``Ref``    -- oracle/_ref/libjpegref.so: the unmodified reference decode path
               compiled from /root/reference/src (exists only where that tree was
               present at build time; prebuilt copy travels to the GPU box).
* ``Oracle`` -- oracle/liboracle.so: the CPU restatement (jpeg_oracle.c).

Both return the same ``Decoded`` record so tests can compare them field by field.
"""
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


@dataclass
class Decoded:
    X: int
    Y: int
    ncs: int
    hsf: list
    vsf: list
    planes: list          # list of (h, w) uint8 arrays, padded to whole MCUs
    rgb: np.ndarray       # (Y, X, 3) uint8, or None
    coefs: np.ndarray     # (nblocks, 64) int16, zigzag order, scan order, or None


class _Out(C.Structure):
    _fields_ = [
        ("X", C.c_int32), ("Y", C.c_int32), ("ncs", C.c_int32),
        ("hsf", C.c_int32 * 4), ("vsf", C.c_int32 * 4),
        ("plane_w", C.c_int32 * 4), ("plane_h", C.c_int32 * 4),
        ("plane", C.POINTER(C.c_uint8) * 4),
        ("rgb", C.POINTER(C.c_uint8)),
        ("pitch", C.c_int32),
        ("nblocks", C.c_int64),
        ("coefs", C.POINTER(C.c_int16)),
    ]


def _unpack(o, want_tap, want_rgb):
    planes = []
    for i in range(o.ncs):
        w, h = o.plane_w[i], o.plane_h[i]
        planes.append(np.ctypeslib.as_array(o.plane[i], shape=(h, w)).copy())
    rgb = None
    if want_rgb:
        rgb = np.ctypeslib.as_array(o.rgb, shape=(o.Y, o.pitch)).copy()[:, : 3 * o.X].reshape(o.Y, o.X, 3)
    coefs = None
    if want_tap:
        if o.nblocks:
            coefs = np.ctypeslib.as_array(o.coefs, shape=(o.nblocks, 64)).copy()
        else:
            coefs = np.zeros((0, 64), np.int16)
    return Decoded(o.X, o.Y, o.ncs, list(o.hsf)[: o.ncs], list(o.vsf)[: o.ncs], planes, rgb, coefs)


class _Lib:
    prefix = ""
    path = ""

    def __init__(self):
        self.lib = C.CDLL(self.path)
        p = self.prefix
        self._decode = getattr(self.lib, p + "_decode")
        self._decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(_Out)]
        self._decode.restype = C.c_int
        self._free = getattr(self.lib, p + "_free")
        self._free.argtypes = [C.POINTER(_Out)]
        self._time = getattr(self.lib, p + "_time_decode")
        self._time.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
        self._time.restype = C.c_double
        self._dqidct = getattr(self.lib, p + "_dequant_idct")
        self._dqidct.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self._row = getattr(self.lib, p + "_ycbcr_row")
        self._row.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self._row.restype = C.c_int

    @classmethod
    def available(cls):
        return os.path.exists(cls.path)

    def decode(self, data: bytes, want_tap=True, want_rgb=True) -> Decoded:
        o = _Out()
        rc = self._decode(data, len(data), int(want_tap), int(want_rgb), C.byref(o))
        if rc != 0:
            raise RuntimeError(f"{self.prefix}_decode failed")
        try:
            return _unpack(o, want_tap, want_rgb)
        finally:
            self._free(C.byref(o))

    def time_decode(self, data: bytes, iters: int) -> float:
        return float(self._time(data, len(data), iters))

    def dequant_idct(self, zz: np.ndarray, qt: np.ndarray) -> np.ndarray:
        zz = np.ascontiguousarray(zz, np.int16)
        qt = np.ascontiguousarray(qt, np.uint16)
        out = np.empty(64, np.int16)
        self._dqidct(zz.ctypes.data, qt.ctypes.data, out.ctypes.data)
        return out

    def ycbcr_row(self, y, cb, cr) -> np.ndarray:
        y = np.ascontiguousarray(y, np.uint8)
        cb = np.ascontiguousarray(cb, np.uint8)
        cr = np.ascontiguousarray(cr, np.uint8)
        n = y.size
        out = np.zeros(3 * n + 16, np.uint8)
        rc = self._row(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, n, out.ctypes.data)
        assert rc == 0
        return out[: 3 * n].reshape(n, 3)


CHECK_DTYPE = np.dtype([("pos", "<i8"), ("off", "<i4"), ("ret", "<i4"), ("blocks", "<i8"), ("pos_after", "<i8")])


@dataclass
class Traced:
    """What the compiled reference did in a child process of its own, read from memory shared
    with it: complete even when the child died."""
    signal: int            # the signal that ended the child, 0 if it exited
    exit_code: int         # its exit status (when signal == 0)
    finished: bool         # decode_jpeg_buffer returned
    rc: int                # ... with this value
    coefs: np.ndarray      # (n, 64) int16: every block the coefficient tap saw, zigzag order
    checks: np.ndarray     # CHECK_DTYPE, one record per MCU: cursor at check_marker, its answer, blocks tapped before,
                           # cursor after next_byte_restart_marker (-1: not called)


class Ref(_Lib):
    prefix = "ref"
    path = os.path.join(HERE, "_ref", "libjpegref.so")

    def decode_traced(self, data: bytes, cap_blocks: int, cap_checks: int) -> Traced:
        """decode_jpeg_buffer in a forked child with the coefficient tap and the restart
        checks recorded in a MAP_SHARED buffer (ref_harness.c: ref_decode_traced)."""
        import mmap
        self.lib.ref_trace_bytes.restype = C.c_size_t
        self.lib.ref_trace_bytes.argtypes = [C.c_int64, C.c_int64]
        self.lib.ref_decode_traced.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        nbytes = self.lib.ref_trace_bytes(cap_blocks, cap_checks)
        shm = mmap.mmap(-1, nbytes)                      # anonymous and shared: survives the child
        head = np.frombuffer(shm, np.int64, 6)
        head[0], head[1] = cap_blocks, cap_checks
        addr = C.addressof(C.c_char.from_buffer(shm))
        pid = os.fork()
        if pid == 0:
            code = 1
            try:
                import faulthandler
                faulthandler.disable()            # the death of this child is the expected outcome, not news
                devnull = os.open(os.devnull, os.O_WRONLY)
                os.dup2(devnull, 1)
                os.dup2(devnull, 2)
                code = 0 if self.lib.ref_decode_traced(data, len(data), addr) == 0 else 2
            finally:
                os._exit(code)
        _, status = os.waitpid(pid, 0)
        n_blocks, n_checks, finished, rc = (int(v) for v in head[2:6])
        coefs = np.frombuffer(shm, np.int16, n_blocks * 64, 48).reshape(n_blocks, 64).copy()
        checks = np.frombuffer(shm, CHECK_DTYPE, n_checks, 48 + cap_blocks * 128).copy()
        del head
        return Traced(os.WTERMSIG(status) if os.WIFSIGNALED(status) else 0,
                      os.WEXITSTATUS(status) if os.WIFEXITED(status) else -1, bool(finished), rc, coefs, checks)


class Oracle(_Lib):
    prefix = "orc"
    path = os.path.join(HERE, "liboracle.so")
