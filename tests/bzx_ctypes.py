"""ctypes bindings used by the tests, bench.py and __graft_entry__.py.

BzxLib   -> the product C ABI (include/bzx.h), bzip2-rust_amd/libbzx.so  (HIP, needs a GPU)
Oracle   -> the CPU oracle (oracle/bzx_oracle.h), oracle/libbzx_oracle.so (checker only)
"""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("BZX_LIB", os.path.join(ROOT, "bzip2-rust_amd", "libbzx.so"))   # BZX_LIB: probe builds
EMU_PATH = os.path.join(ROOT, "tests", "emu", "libbzx_emu.so")
ORACLE_PATH = os.path.join(ROOT, "oracle", "libbzx_oracle.so")

u8p = C.POINTER(C.c_uint8)


class BzxStats(C.Structure):
    _fields_ = [("nblk", C.c_uint32), ("n_periodic", C.c_uint32), ("raw_bytes", C.c_uint64),
                ("rle1_bytes", C.c_uint64), ("mtf_symbols", C.c_uint64), ("out_bits", C.c_uint64),
                ("ms_split", C.c_float), ("ms_bwt", C.c_float), ("ms_mtf", C.c_float),
                ("ms_huffman", C.c_float), ("ms_emit", C.c_float), ("ms_total", C.c_float),
                ("bwt_launches", C.c_uint32), ("n_redo", C.c_uint32), ("n_buckets", C.c_uint32),
                ("ms_bwt_split", C.c_float), ("ms_bwt_sort", C.c_float), ("ms_bwt_general", C.c_float),
                ("n_open_buckets", C.c_uint32), ("n_open_left", C.c_uint32), ("n_resume_left", C.c_uint32),
                ("ms_bwt_rank", C.c_float), ("n_from_scratch", C.c_uint32), ("n_unsorted", C.c_uint32)]


class BzxBlockInfo(C.Structure):
    _fields_ = [("n", C.c_uint32), ("crc", C.c_uint32), ("orig_ptr", C.c_uint32), ("periodic", C.c_uint32),
                ("n_in_use", C.c_uint32), ("n_mtf", C.c_uint32), ("n_tables", C.c_uint32), ("n_selectors", C.c_uint32),
                ("bits_symbol_map", C.c_uint32), ("bits_selectors", C.c_uint32), ("bits_tables", C.c_uint32),
                ("bits_payload", C.c_uint32), ("bits", C.c_uint64)]


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
        L.bzx_stage_mtf.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint32),
                                    C.c_void_p, C.c_void_p]
        L.bzx_stage_huffman.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p,
                                        C.c_void_p]
        L.bzx_compress_block.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_char_p, C.c_size_t,
                                         C.POINTER(C.c_size_t), C.POINTER(C.c_uint8)]
        L.bzx_compress_blocks.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
        L.bzx_get_stats.argtypes = [C.c_void_p, C.POINTER(BzxStats)]
        L.bzx_compress_buffer.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t,
                                          C.POINTER(C.c_size_t)]
        L.bzx_compress_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                          C.POINTER(C.c_size_t)]
        L.bzx_split_rle1.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.bzx_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
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


    def stage_mtf(self, bwt: bytes):
        n = len(bwt)
        mtfv = (C.c_uint16 * (n + 2))()
        n_mtf = C.c_uint32()
        freq = (C.c_uint32 * 258)()
        in_use = (C.c_uint8 * 256)()
        self._check(self.lib.bzx_stage_mtf(self.ctx, bwt, n, mtfv, C.byref(n_mtf), freq, in_use))
        return list(mtfv[:n_mtf.value]), list(freq), bytes(in_use)

    def stage_huffman(self, mtfv, freq, alpha):
        n_mtf = len(mtfv)
        arr = (C.c_uint16 * n_mtf)(*mtfv)
        f = (C.c_uint32 * 258)(*freq)
        ng = C.c_uint32()
        ns = C.c_uint32()
        sel = (C.c_uint8 * 18002)()
        ln = (C.c_uint8 * (6 * 258))()
        code = (C.c_uint32 * (6 * 258))()
        self._check(self.lib.bzx_stage_huffman(self.ctx, arr, n_mtf, f, alpha, C.byref(ng), C.byref(ns), sel, ln, code))
        lens = [list(ln[t * 258:t * 258 + alpha]) for t in range(ng.value)]
        codes = [list(code[t * 258:t * 258 + alpha]) for t in range(ng.value)]
        return ng.value, list(sel[:ns.value]), lens, codes

    def compress_blocks(self, blocks, crcs):
        """blocks: list of bytes (RLE1'd), crcs: list of int -> list of (image bytes, pad_bits)."""
        nb = len(blocks)
        caps = [len(b) + len(b) // 50 + 1024 for b in blocks]
        outs = [C.create_string_buffer(c) for c in caps]
        p_in = (C.c_char_p * nb)(*blocks)
        p_ns = (C.c_size_t * nb)(*[len(b) for b in blocks])
        p_crc = (C.c_uint32 * nb)(*crcs)
        p_out = (C.c_void_p * nb)(*[C.addressof(o) for o in outs])
        p_cap = (C.c_size_t * nb)(*caps)
        p_len = (C.c_size_t * nb)()
        p_pad = (C.c_uint8 * nb)()
        self._check(self.lib.bzx_compress_blocks(self.ctx, nb, p_in, p_ns, p_crc, p_out, p_cap, p_len, p_pad))
        return [(outs[i].raw[:p_len[i]], p_pad[i]) for i in range(nb)]

    def compress_block(self, blk: bytes, crc: int):
        """The per-block entry point itself (compress_block.rs:24); safe to call from several threads at once."""
        cap = len(blk) + len(blk) // 50 + 1024
        out = C.create_string_buffer(cap)
        ol, pad = C.c_size_t(), C.c_uint8()
        self.lib.bzx_compress_block.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t,
                                                C.POINTER(C.c_size_t), C.POINTER(C.c_uint8)]
        self._check(self.lib.bzx_compress_block(self.ctx, blk, len(blk), crc, out, cap, C.byref(ol), C.byref(pad)))
        return out.raw[:ol.value], pad.value

    def compress_buffer(self, data: bytes, level=9):
        cap = len(data) + len(data) // 50 + 4096
        out = C.create_string_buffer(cap)
        ol = C.c_size_t()
        self._check(self.lib.bzx_compress_buffer(self.ctx, data, len(data), level, out, cap, C.byref(ol)))
        return out.raw[:ol.value]

    def compress_device(self, d_raw_ptr, nbytes, level, d_out_ptr, cap):
        """Device pointers in, stream length out (nothing but the length crosses PCIe)."""
        ol = C.c_size_t()
        self._check(self.lib.bzx_compress_device(self.ctx, d_raw_ptr, nbytes, level, d_out_ptr, cap, C.byref(ol)))
        return ol.value

    def split_rle1(self, data: bytes, level=9):
        nmax = 100000 * level - 19
        cap = (len(data) + len(data) // 4) // nmax + 2
        slabs = C.create_string_buffer(cap * 900000)
        ns = (C.c_uint32 * cap)()
        crcs = (C.c_uint32 * cap)()
        nb = C.c_uint32()
        self._check(self.lib.bzx_split_rle1(self.ctx, data, len(data), level, slabs, cap, ns, crcs, C.byref(nb)))
        return [(slabs.raw[b * 900000:b * 900000 + ns[b]], crcs[b]) for b in range(nb.value)]

    def cstream_compress(self, data: bytes, level=9, chunk=1 << 20, max_chunk=None, pinned=False):
        """bzx_cstream_*: feed `data` in pieces of `chunk` bytes (an int, or a list of piece lengths); returns the .bz2."""
        L = self.lib
        L.bzx_cstream_begin.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
        L.bzx_cstream_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                       C.POINTER(C.c_size_t)]
        L.bzx_cstream_end.argtypes = [C.c_void_p]
        pieces = chunk if isinstance(chunk, (list, tuple)) else None
        mc = max_chunk or (max(pieces) if pieces else chunk)
        s = C.c_void_p()
        self._check(L.bzx_cstream_begin(self.ctx, level, mc, C.byref(s)))
        cap = len(data) + len(data) // 50 + 4096
        out = C.create_string_buffer(cap)
        src = C.create_string_buffer(data, len(data)) if data else C.create_string_buffer(1)
        produced = C.c_size_t()
        try:
            off, i, last = 0, 0, 0
            while True:
                n = min(len(data) - off, pieces[i % len(pieces)] if pieces else chunk)
                fin = off + n >= len(data)
                self._check(L.bzx_cstream_feed(s, C.addressof(src) + off, n, int(fin), out, cap, C.byref(produced)))
                assert produced.value >= last
                last = produced.value
                off += n
                i += 1
                if fin:
                    break
        finally:
            L.bzx_cstream_end(s)
        return out.raw[:produced.value]

    def split_rle1_chunks(self, data: bytes, level=9, chunk=1 << 16):
        """bzx_split_rle1_chunk over pieces of `chunk` bytes -> list of (block bytes, crc), as split_rle1."""
        L = self.lib
        L.bzx_split_rle1_chunk.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_uint32,
                                           C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        capb = 8
        buf = C.create_string_buffer(capb * 900000)
        ns, crcs, nb = (C.c_uint32 * capb)(), (C.c_uint32 * capb)(), C.c_uint32()
        out, off = [], 0
        while True:
            n = min(chunk, len(data) - off)
            fin = off + n >= len(data)
            self._check(L.bzx_split_rle1_chunk(self.ctx, data[off:off + n], n, level, int(fin), buf, capb, ns, crcs, C.byref(nb)))
            for b in range(nb.value):
                out.append((buf.raw[b * 900000:b * 900000 + ns[b]], crcs[b]))
            off += n
            if fin:
                return out

    def decompress_buffer(self, z: bytes, cap=None):
        """bzx_decompress_buffer: one .bz2 stream -> raw bytes (block and combined CRCs verified on the device)."""
        L = self.lib
        L.bzx_decompress_buffer.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        cap = cap if cap is not None else max(1 << 16, len(z) * 8)
        while True:
            out = C.create_string_buffer(cap)
            ol = C.c_size_t()
            rc = L.bzx_decompress_buffer(self.ctx, z, len(z), out, cap, C.byref(ol))
            if rc == -4 and ol.value > cap:         # BZX_E_OUTBUF: the needed size is reported
                cap = ol.value
                continue
            self._check(rc)
            return out.raw[:ol.value]

    def block_info(self, i):
        bi = BzxBlockInfo()
        self.lib.bzx_get_block_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(BzxBlockInfo)]
        self._check(self.lib.bzx_get_block_info(self.ctx, i, C.byref(bi)))
        return bi

    def stats(self):
        st = BzxStats()
        self._check(self.lib.bzx_get_stats(self.ctx, C.byref(st)))
        return st


class OracleHuff(C.Structure):
    _fields_ = [("n_groups", C.c_int32), ("n_selectors", C.c_int32), ("selector", C.c_uint8 * 18002),
                ("selector_mtf", C.c_uint8 * 18002), ("len", (C.c_uint8 * 258) * 6), ("code", (C.c_int32 * 258) * 6)]


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

    def mtf(self, bwt: bytes):
        n = len(bwt)
        mtfv = (C.c_uint16 * (n + 2))()
        freq = (C.c_int32 * 258)()
        in_use = (C.c_uint8 * 256)()
        niu = C.c_int32()
        self.lib.bzo_mtf_rle2.restype = C.c_int32
        m = self.lib.bzo_mtf_rle2(bwt, n, mtfv, freq, in_use, C.byref(niu))
        return list(mtfv[:m]), list(freq), bytes(in_use), niu.value

    def huff(self, mtfv, freq, alpha):
        T = OracleHuff()
        arr = (C.c_uint16 * len(mtfv))(*mtfv)
        f = (C.c_int32 * 258)(*freq)
        self.lib.bzo_huff_optimise(arr, len(mtfv), f, alpha, C.byref(T))
        lens = [list(T.len[t][:alpha]) for t in range(T.n_groups)]
        codes = [list(T.code[t][:alpha]) for t in range(T.n_groups)]
        return T.n_groups, list(T.selector[:T.n_selectors]), lens, codes

    def split_rle1(self, data: bytes, level=9):
        self.lib.bzo_rle1_block.restype = C.c_size_t
        self.lib.bzo_rle1_block.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_int, C.c_void_p,
                                            C.c_char_p, C.POINTER(C.c_uint32)]
        st = (C.c_uint32 * 2)(256, 0)
        pos = C.c_size_t(0)
        blk = C.create_string_buffer(100000 * level + 16)
        out = []
        while pos.value < len(data) or st[0] < 256:
            crc = C.c_uint32()
            n = self.lib.bzo_rle1_block(data, len(data), C.byref(pos), level, st, blk, C.byref(crc))
            if n == 0:
                break
            out.append((blk.raw[:n], crc.value))
        return out

    def crc32(self, data: bytes):
        self.lib.bzo_crc32.restype = C.c_uint32
        self.lib.bzo_crc32.argtypes = [C.c_char_p, C.c_size_t]
        return self.lib.bzo_crc32(data, len(data))

    def compress_block(self, blk: bytes, crc: int):
        cap = len(blk) + len(blk) // 50 + 1024
        out = C.create_string_buffer(cap)
        ol = C.c_size_t()
        pad = C.c_uint8()
        self.lib.bzo_compress_block.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_char_p, C.c_size_t,
                                                C.POINTER(C.c_size_t), C.POINTER(C.c_uint8)]
        rc = self.lib.bzo_compress_block(blk, len(blk), crc, out, cap, C.byref(ol), C.byref(pad))
        assert rc == 0
        return out.raw[:ol.value], pad.value

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

    def compress_mt(self, data: bytes, level=9, threads=None):
        """Same stream as compress(), one block per worker thread (oracle_driver.c; compress.rs:125-132)."""
        L = self.lib
        L.bzo_compress_buffer_mt.restype = C.c_size_t
        L.bzo_compress_buffer_mt.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_char_p, C.c_size_t,
                                             C.POINTER(C.c_int32)]
        cap = len(data) + len(data) // 50 + 4096
        out = C.create_string_buffer(cap)
        nb = C.c_int32()
        n = L.bzo_compress_buffer_mt(data, len(data), level, threads or min(16, os.cpu_count() or 1), out, cap, C.byref(nb))
        if n == 0 and len(data):
            raise RuntimeError("oracle compress_mt overflow")
        return out.raw[:n]

    def compress(self, data: bytes, level=9):
        cap = len(data) + len(data) // 50 + 4096
        out = C.create_string_buffer(cap)
        nb = C.c_int32()
        n = self.lib.bzo_compress_buffer(data, len(data), level, out, cap, C.byref(nb))
        return out.raw[:n], nb.value
