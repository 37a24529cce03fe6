"""ctypes bindings used by the tests, bench.py and __graft_entry__.py.

BzxLib   -> the product C ABI (include/bzx.h), bzip2-rust_amd/libbzx.so  (HIP, needs a GPU)
Oracle   -> the CPU oracle (oracle/bzx_oracle.h), oracle/libbzx_oracle.so (checker only)
"""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "bzip2-rust_amd", "libbzx.so")
EMU_PATH = os.path.join(ROOT, "tests", "emu", "libbzx_emu.so")
ORACLE_PATH = os.path.join(ROOT, "oracle", "libbzx_oracle.so")

u8p = C.POINTER(C.c_uint8)


class BzxStats(C.Structure):
    _fields_ = [("nblk", C.c_uint32), ("n_periodic", C.c_uint32), ("raw_bytes", C.c_uint64),
                ("rle1_bytes", C.c_uint64), ("mtf_symbols", C.c_uint64), ("out_bits", C.c_uint64),
                ("ms_split", C.c_float), ("ms_bwt", C.c_float), ("ms_mtf", C.c_float),
                ("ms_huffman", C.c_float), ("ms_emit", C.c_float), ("ms_total", C.c_float)]


class BzxError(RuntimeError):
    pass


class BzxLib:
    """Thin wrapper over the C ABI; raises BzxError on any non-zero return."""

    def __init__(self, path=LIB_PATH, device=0, max_blocks=16):
        if not os.path.exists(path):
            raise BzxError(f"{path} missing: build it with __graft_entry__.build() (no CPU fallback exists)")
        self.lib = L = C.CDLL(path)
        L.bzx_version.restype = C.c_char_p
        L.bzx_strerror.restype = C.c_char_p
        L.bzx_strerror.argtypes = [C.c_int]
        L.bzx_last_error.restype = C.c_char_p
        L.bzx_last_error.argtypes = [C.c_void_p]
        L.bzx_ctx_create.argtypes = [C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
        L.bzx_ctx_destroy.argtypes = [C.c_void_p]
        L.bzx_stage_bwt.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_uint32)]
        self.ctx = C.c_void_p()
        self._check(L.bzx_ctx_create(device, max_blocks, C.byref(self.ctx)))

    def _check(self, rc):
        if rc != 0:
            msg = self.lib.bzx_strerror(rc).decode()
            if self.ctx:
                msg += ": " + self.lib.bzx_last_error(self.ctx).decode()
            raise BzxError(f"bzx error {rc}: {msg}")

    def close(self):
        if self.ctx:
            self.lib.bzx_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def stage_bwt(self, blk: bytes):
        n = len(blk)
        out = C.create_string_buffer(n)
        orig = C.c_uint32()
        status = C.c_uint32()
        self._check(self.lib.bzx_stage_bwt(self.ctx, blk, n, out, C.byref(orig), C.byref(status)))
        return out.raw[:n], orig.value, status.value


class Oracle:
    def __init__(self, path=ORACLE_PATH):
        self.lib = L = C.CDLL(path)
        L.bzo_bwt.restype = C.c_int32
        L.bzo_bwt.argtypes = [C.c_char_p, C.c_int32, C.c_char_p, C.c_void_p]
        L.bzo_synthtext.argtypes = [C.c_uint64, C.c_char_p, C.c_size_t]
        L.bzo_xorshift_bytes.argtypes = [C.c_uint64, C.c_char_p, C.c_size_t]
        L.bzo_compress_buffer.restype = C.c_size_t
        L.bzo_compress_buffer.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t,
                                          C.POINTER(C.c_int32)]

    def bwt(self, blk: bytes):
        n = len(blk)
        out = C.create_string_buffer(n)
        orig = self.lib.bzo_bwt(blk, n, out, None)
        return out.raw[:n], orig

    def synthtext(self, n, seed=0x9E3779B97F4A7C15):
        b = C.create_string_buffer(n)
        self.lib.bzo_synthtext(seed, b, n)
        return b.raw[:n]

    def randbytes(self, n, seed=0xD1B54A32D192ED03):
        b = C.create_string_buffer(n)
        self.lib.bzo_xorshift_bytes(seed, b, n)
        return b.raw[:n]

    def compress(self, data: bytes, level=9):
        cap = len(data) + len(data) // 50 + 4096
        out = C.create_string_buffer(cap)
        nb = C.c_int32()
        n = self.lib.bzo_compress_buffer(data, len(data), level, out, cap, C.byref(nb))
        return out.raw[:n], nb.value
