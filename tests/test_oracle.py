"""CPU tests (-m "not gpu"): pin the oracle.

The Rust reference cannot be built here and its own tests pin only bit packing
(src/bitstream/bitpacker.rs:118-166) and the symbol map (src/tools/symbol_map.rs:45-59).  Those
vectors are replayed below; everything else is pinned by whole-stream byte equality with libbz2
1.0.8 (python bz2 -- the library BASELINE.json's metric names) live and through the committed
fixtures in tests/golden/streams.json."""
import bz2
import ctypes as C
import hashlib
import json
import os
import random
import subprocess

import pytest

from gen_golden import make_input

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "streams.json")))


def test_known_answer_streams(oracle):
    for k in ("empty_l9", "silly_l9"):
        v = GOLDEN["kats"][k]
        out, _ = oracle.compress(bytes.fromhex(v["input_hex"]), v["level"])
        assert out.hex() == v["bz2_hex"]
    # SURVEY.md 3.4 inline vectors
    assert GOLDEN["kats"]["empty_l9"]["bz2_hex"] == "425a683917724538509000000000"
    assert GOLDEN["kats"]["silly_l9"]["bz2_hex"].startswith("425a683931415926535955066e80")


def test_reference_bitpacker_vectors(oracle):
    """bitpacker.rs:118-166: out16_test, out24_and_loc_test, out24_short_test, out32_test."""
    L = oracle.lib

    class BP(C.Structure):
        _fields_ = [("out", C.c_void_p), ("cap", C.c_size_t), ("len", C.c_size_t), ("queue", C.c_uint64),
                    ("q_bits", C.c_int), ("overflow", C.c_int)]

    for v in GOLDEN["kats"]["bitpacker"]:
        buf = C.create_string_buffer(64)
        bp = BP()
        L.bzo_bp_init(C.byref(bp), buf, 64)
        for op, arg in v["ops"]:
            if op == "out16":
                L.bzo_bp_out16(C.byref(bp), C.c_uint16(arg))
            elif op == "out24":
                L.bzo_bp_out24(C.byref(bp), C.c_uint32(arg))
            elif op == "out32":
                L.bzo_bp_out32(C.byref(bp), C.c_uint32(arg))
            elif op == "flush":
                L.bzo_bp_flush(C.byref(bp))
        L.bzo_bp_flush(C.byref(bp))
        assert buf.raw[:bp.len].hex() == v["bytes_hex"], v["name"]


def test_reference_symbol_map_vectors(oracle):
    """symbol_map.rs:45-59 run in the encode direction."""
    for k in ("symbol_map_silly", "symbol_map_full"):
        v = GOLDEN["kats"][k]
        data = bytes.fromhex(v["input_hex"])
        in_use = (C.c_uint8 * 256)()
        for b in data:
            in_use[b] = 1
        words = (C.c_uint16 * 17)()
        n = oracle.lib.bzo_symbol_map(in_use, words)
        assert list(words[:n]) == v["words"]


@pytest.mark.parametrize("name", sorted(GOLDEN["streams"].keys()))
def test_golden_streams(oracle, name):
    g = GOLDEN["streams"][name]
    data = make_input(oracle, g["input"])
    assert len(data) == g["raw_len"] and hashlib.sha256(data).hexdigest() == g["raw_sha256"]
    # (the full-size BASELINE configs go through the multi-threaded driver: same bytes, minutes less)
    out = oracle.compress(data, g["level"])[0] if len(data) <= (64 << 20) else oracle.compress_mt(data, g["level"], 8)
    assert len(out) == g["bz2_len"]
    assert hashlib.sha256(out).hexdigest() == g["bz2_sha256"]


def test_periodic_orig_ptr_fixture(oracle):
    """origPtr of blocks that are a power u^k is whatever libbz2's sorter leaves; the committed sweep
    (tests/golden/periodic.json, read out of libbz2 1.0.8 streams by tests/gen_golden.py) pins the oracle's block
    sorter, which the device's periodic path is compared with."""
    import json
    import os
    from gen_golden import periodic_unit
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "periodic.json")))
    assert len(fx["blocks"]) >= 20
    for g in fx["blocks"]:
        raw = periodic_unit(g["unit"]) * g["k"]
        image = oracle.split_rle1(raw, 9)[0][0]
        assert len(image) == g["n"] and hashlib.sha256(image).hexdigest() == g["sha256"]
        assert oracle.bwt(image)[1] == g["orig_ptr"], (g["unit"], g["k"])


def test_live_libbz2_edge_cases(oracle):
    """Every run length 1..600, block-boundary cases of the split rule, periodic blocks (D6)."""
    rnd = random.Random(1)
    cases = []
    for n in list(range(1, 12)) + [254, 255, 256, 257, 258, 259, 260, 509, 510, 511, 600]:
        cases.append((b"x" * n, 9))
        cases.append((b"b" + b"a" * n + b"c", 9))
    for n in (99981, 99982, 99983, 199962, 199963, 199964):
        cases.append((oracle.synthtext(n), 1))
    cases += [(b"ab" * 49989 + b"c" * 259, 1), (b"ab" * 49990 + b"c" * 300 + b"xyz", 1)]
    # periodic blocks: origPtr depends on libbz2's sorter internals
    cases += [(b"\0\0\0\1" * 3000, 9), (b"abcabcd" * 15, 9), (b"abcabcd" * 20000, 9), (b"ab" * 30000, 9),
              (rnd.randbytes(5000) * 4, 9), (rnd.randbytes(45000) * 2, 9), (rnd.randbytes(300) * 300, 9),
              (b"\0" * 5_000_000, 9), (bytes(range(256)) * 40, 9)]
    cases += [(rnd.randbytes(n), 9) for n in (10, 100, 1000, 9999, 10000, 10001, 50000)]
    cases += [(bytes(rnd.choice(b"abc") for _ in range(n)), 9) for n in (10, 1000, 10001)]
    for data, level in cases:
        out, _ = oracle.compress(data, level)
        assert out == bz2.compress(data, level), (len(data), level)


def test_bzip2_cli_agrees(oracle):
    """/usr/bin/bzip2 (the C program itself) on the split-rule edge: same bytes as the oracle."""
    if not os.path.exists("/usr/bin/bzip2"):
        pytest.skip("bzip2 CLI not installed")
    for data, level in ((oracle.synthtext(99982), 1), (oracle.synthtext(250000), 1), (b"\0" * 3_000_000, 9)):
        cli = subprocess.run(["/usr/bin/bzip2", f"-{level}", "-c"], input=data, capture_output=True, check=True).stdout
        assert oracle.compress(data, level)[0] == cli


def test_multithreaded_driver_matches(oracle):
    """compress.rs:125-132 shape: one block per worker, ordered assembly."""
    L = oracle.lib
    L.bzo_compress_buffer_mt.restype = C.c_size_t
    L.bzo_compress_buffer_mt.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_char_p, C.c_size_t,
                                         C.POINTER(C.c_int32)]
    data = oracle.synthtext(1 << 20)
    cap = len(data) + 4096
    out = C.create_string_buffer(cap)
    nb = C.c_int32()
    n = L.bzo_compress_buffer_mt(data, len(data), 1, 4, out, cap, C.byref(nb))
    assert out.raw[:n] == bz2.compress(data, 1) and nb.value == 11


def test_stage_functions_consistent(oracle):
    """Stage-level oracle entry points reproduce compress_block's bytes when chained (used by -m gpu tests)."""
    blk = oracle.synthtext(30000)
    L, orig = oracle.bwt(blk)
    # BWT definition (bwt_sort.rs:45-57): sorted rotations, preceding byte
    n = len(blk)
    rots = sorted(range(n), key=lambda i: blk[i:] + blk[:i])
    assert bytes(blk[i - 1] for i in rots) == L and rots.index(0) == orig
    mtfv, freq, in_use, niu = oracle.mtf(L)
    assert mtfv[-1] == niu + 1 and sum(freq) == len(mtfv)
    ng, sel, lens, codes = oracle.huff(mtfv, freq, niu + 2)
    assert ng == 6 and len(sel) == (len(mtfv) + 49) // 50 and max(max(l) for l in lens) <= 17
