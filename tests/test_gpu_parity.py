"""-m gpu: parity of every device stage, of the per-block C ABI and of the whole-stream C ABI
against the oracle (oracle/), libbz2 (python bz2) and the golden fixtures.  Bit-exact: this is
integer/byte work, there is no tolerance."""
import bz2
import hashlib
import json
import os
import random

import pytest

pytestmark = pytest.mark.gpu
GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "streams.json")))


def _cases(oracle):
    rnd = random.Random(11)
    yield "one", b"a"
    yield "two_equal", b"aa"                       # periodic
    yield "banana", b"banana"
    yield "silly", b"Making a silly test."         # reference KAT string (symbol_map.rs:45-59)
    yield "text20k", oracle.synthtext(20000)
    yield "lowent", bytes(rnd.choice(b"ab") for _ in range(60000))
    yield "bytes256", rnd.randbytes(100000)
    yield "zeros_rle", b"\0\0\0\0\xfb" * 2000 + b"\0\0\0\0\x10"    # RLE1 image of zeros, not periodic
    yield "text900k", oracle.synthtext(899981)
    yield "rand900k", oracle.randbytes(899981)
    yield "maxblock", oracle.synthtext(900000)     # largest block the ABI accepts
    yield "mix", b"\0" * 300000 + oracle.synthtext(300000) + b"\xff" * 299000


def test_stage_bwt(bzx, oracle):
    """bwt_encode contract (bwt_sort.rs:27-58): last column and origPtr."""
    for name, blk in _cases(oracle):
        L, orig, status = bzx.stage_bwt(blk)
        Lo, oo = oracle.bwt(blk)
        assert L == Lo, name
        assert orig == oo, name


def test_stage_mtf(bzx, oracle):
    """rle2_mtf_encode contract (rle2_mtf.rs:23-177): symbols, histogram, bytes in use."""
    for name, blk in _cases(oracle):
        L, _ = oracle.bwt(blk)
        mt, fr, iu = bzx.stage_mtf(L)
        mo, fo, iuo, _ = oracle.mtf(L)
        assert mt == mo, name
        assert fr == fo, name
        assert iu == iuo, name


def test_stage_huffman(bzx, oracle):
    """huf_encode table optimisation (huffman.rs:87-374): tables, selectors, lengths, codes."""
    for name, blk in _cases(oracle):
        L, _ = oracle.bwt(blk)
        mo, fo, _, niu = oracle.mtf(L)
        got = bzx.stage_huffman(mo, fo, niu + 2)
        want = oracle.huff(mo, fo, niu + 2)
        assert got == want, name


def test_compress_block(bzx, oracle):
    """compress_block (compress_block.rs:24-67): byte-aligned image + pad bits."""
    for name, blk in _cases(oracle):
        crc = oracle.crc32(blk)
        assert bzx.compress_block(blk, crc) == oracle.compress_block(blk, crc), name


def test_compress_block_from_many_threads(bzx, oracle):
    """The reference's calling pattern (compress.rs:125-132): compress_block from every worker thread at once, one
    context.  The library batches concurrent calls; every caller must get its own block image."""
    import threading
    rnd = random.Random(23)
    blocks = [oracle.synthtext(200000 + 1000 * i) if i % 4 else rnd.randbytes(150000 + 777 * i) for i in range(48)]
    crcs = [oracle.crc32(b) for b in blocks]
    got = [None] * len(blocks)
    biggest = [0]

    def work(i):
        got[i] = bzx.compress_block(blocks[i], crcs[i])
        biggest[0] = max(biggest[0], bzx.stats().nblk)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(blocks))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    for i, b in enumerate(blocks):
        assert got[i] == oracle.compress_block(b, crcs[i]), i
    assert biggest[0] > 1          # concurrent calls really shared a device batch


def test_compress_blocks_batched(bzx, oracle):
    """Batched form (compress.rs:125-132): many ragged blocks at once, more blocks than CUs*slots is fine."""
    rnd = random.Random(5)
    blocks = [oracle.synthtext(rnd.randint(1, 30000), seed=1000 + i) for i in range(40)]
    blocks += [rnd.randbytes(rnd.randint(1, 5000)) for _ in range(20)]
    crcs = [oracle.crc32(b) for b in blocks]
    got = bzx.compress_blocks(blocks, crcs)
    for i, b in enumerate(blocks):
        assert got[i] == oracle.compress_block(b, crcs[i]), i


def test_split_rle1(bzx, oracle):
    """RLE1Block (rle1.rs:33-263) with libbz2's split rule + do_crc (crc.rs:15-22)."""
    rnd = random.Random(3)

    def runs(n, maxrun, alphabet):
        out = bytearray()
        while len(out) < n:
            out += bytes([rnd.choice(alphabet)]) * rnd.randint(1, maxrun)
        return bytes(out[:n])

    cases = [(oracle.synthtext(250000), 1), (runs(300000, 600, b"ab\0"), 1), (b"\0" * 6_000_000, 1),
             (runs(250000, 7, b"abc"), 1), (b"aaaaab", 9), (b"x" * 255 + b"y" * 256 + b"z" * 4, 9),
             (oracle.synthtext(99982), 1), (b"ab" * 49989 + b"c" * 259, 1), (oracle.synthtext(2_000_000), 9),
             (runs(3_000_000, 300, bytes(range(256))), 9)]
    for data, level in cases:
        assert bzx.split_rle1(data, level) == oracle.split_rle1(data, level)


@pytest.mark.parametrize("name", sorted(GOLDEN["streams"].keys()))
def test_stream_golden(bzx, oracle, name):
    """Whole buffer -> .bz2 (compress.rs:40-136) against the committed libbz2 fixtures and live libbz2."""
    from gen_golden import make_input
    g = GOLDEN["streams"][name]
    data = make_input(oracle, g["input"])
    out = bzx.compress_buffer(data, g["level"])
    assert len(out) == g["bz2_len"]
    assert hashlib.sha256(out).hexdigest() == g["bz2_sha256"]
    assert bzx.stats().n_unsorted == 0      # no bucket's optimistic initial sort failed its check (bzx_bsort.hip)
    if len(data) <= (8 << 20):
        assert out == bz2.compress(data, g["level"])
    else:
        # BASELINE configs[2] / configs[4] at FULL size (1 GiB text, 256 MiB zeros, 256 MiB random): besides libbz2's
        # committed sha256 above, the oracle's multi-threaded driver must give the same bytes, and libbz2 must decode
        # the head of the device stream back to the head of the input
        assert out == oracle.compress_mt(data, g["level"])
        head = bz2.BZ2Decompressor().decompress(out[:3 << 20], max_length=32 << 20)
        assert len(head) > 0 and data[:len(head)] == head


def test_stream_matches_oracle_and_roundtrips(bzx, oracle):
    """configs[1] + ragged sizes: device stream == oracle stream == libbz2, and libbz2 decodes it."""
    for n, level in ((899981, 9), (1 << 20, 1), (3_000_001, 9), (5, 9), (0, 9)):
        data = oracle.synthtext(n) if n else b""
        out = bzx.compress_buffer(data, level)
        assert out == oracle.compress(data, level)[0]
        assert bz2.decompress(out) == data


def test_bad_arguments(bzx):
    from bzx_ctypes import BzxError
    with pytest.raises(BzxError):
        bzx.stage_bwt(b"")
    with pytest.raises(BzxError):
        bzx.compress_buffer(b"abc", 0)
    with pytest.raises(BzxError):
        bzx.compress_block(b"x" * 900001, 0)


def test_periodic_blocks_stream(bzx, oracle):
    """blocks that are u^k.  The last column is tie-invariant, origPtr must be libbz2's."""
    for data, level in ((b"\0" * 6_000_000, 1),               # RLE1 image (0,0,0,0,251)^k: periodic blocks
                        (b"abcabcd" * 15, 9), (bytes(range(256)) * 40, 9), (b"\0\0\0\1" * 3000, 9)):
        out = bzx.compress_buffer(data, level)
        assert out == bz2.compress(data, level), (len(data), level)
        assert bzx.stats().n_periodic >= 1


PERIODIC = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "periodic.json")))


def test_periodic_orig_ptr_fixture(bzx, oracle):
    """SURVEY.md D6 on the committed (u, k) sweep: for blocks that are a power u^k the last column is tie-invariant
    and origPtr must be the one libbz2 1.0.8 wrote (tests/golden/periodic.json, read out of libbz2's streams)."""
    from gen_golden import periodic_unit
    for g in PERIODIC["blocks"]:
        raw = periodic_unit(g["unit"]) * g["k"]
        image = oracle.split_rle1(raw, 9)[0][0]
        assert len(image) == g["n"] and hashlib.sha256(image).hexdigest() == g["sha256"]
        L, orig, status = bzx.stage_bwt(image)
        assert orig == g["orig_ptr"], (g["unit"], g["k"], orig, g["orig_ptr"])
        assert L == oracle.bwt(image)[0], (g["unit"], g["k"])
        is_power = (image + image).find(image, 1) < len(image)      # the RLE1 image itself is u^k, k > 1
        assert bool(status & 1) == is_power, (g["unit"], g["k"])    # BZX_ST_PERIODIC


def test_stream_assembler_fed_with_device_block_images(bzx, oracle):
    """The reference's structure end to end (compress.rs:66-132): RLE1Block producer -> compress_block on the
    device (batched) -> BitWriter.  Here: bzx_split_rle1 -> bzx_compress_blocks -> bzx_stream_* (host assembler),
    every stage through the C ABI, against libbz2."""
    import ctypes as C
    L = bzx.lib
    L.bzx_stream_begin.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.bzx_stream_append_block.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint8]
    L.bzx_stream_finish.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    L.bzx_stream_free.argtypes = [C.c_void_p]
    data = oracle.synthtext(700_000) + b"\xff" * 70_000 + oracle.randbytes(400_000) + b"ab" * 5
    for level in (1, 3):
        blocks = bzx.split_rle1(data, level)
        assert blocks == oracle.split_rle1(data, level)
        images = bzx.compress_blocks([b for b, _ in blocks], [c for _, c in blocks])
        s = C.c_void_p()
        assert L.bzx_stream_begin(level, C.byref(s)) == 0
        for img, pad in images:
            assert L.bzx_stream_append_block(s, img, len(img), pad) == 0
        p, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        assert L.bzx_stream_finish(s, C.byref(p), C.byref(n)) == 0
        z = bytes(p[:n.value])
        L.bzx_stream_free(s)
        assert z == bz2.compress(data, level), level


def test_chunked_stream_compressor(bzx, oracle):
    """bzx_cstream_* on the device: the input arrives in chunks of 64 KiB ... 64 MiB; chunk borders fall inside runs
    longer than 255, exactly at block-full edges and one byte around them; the stream must be libbz2's, whatever the
    chunking (the splitter carries the pending run and the partial block from call to call)."""
    nmax = 899981
    text = oracle.synthtext(3 * nmax + 12345)
    runs = (b"\xff" * 1000 + b"abc") * 3000 + b"\0" * 2_000_000 + oracle.synthtext(500_000)
    for data, level, chunks in (
            (text, 9, [65536]), (text, 9, [nmax]), (text, 9, [nmax - 1, 1, 1, nmax]), (text, 9, [nmax + 1]),
            (text, 9, [1 << 20, 123]), (runs, 9, [65536]), (runs, 9, [1000 * 700 + 130, 777]), (runs, 1, [99981, 5]),
            (b"\0" * (70 << 20), 9, [1 << 20]), (oracle.randbytes(5 << 20), 5, [(1 << 20) + 1])):
        want = bz2.compress(data, level)
        assert bzx.cstream_compress(data, level, chunks, max_chunk=max(chunks)) == want, (len(data), level, chunks)
    big = oracle.synthtext(150 << 20)
    want = oracle.compress_mt(big, 9)
    assert bzx.cstream_compress(big, 9, 64 << 20) == want
    assert bzx.compress_buffer(big, 9) == want


def test_levels_change_on_one_context(bzx, oracle):
    """bzx_compress_buffer keeps its chunked-stream object in the context and starts the next stream on it at whatever
    level the caller asks for: the object must be provisioned for every level (blocks per chunk are most numerous at
    level 1: 16 MiB is 168 blocks there, 19 at level 9; the withheld raw tail of a run-heavy chunk is longest at 9)."""
    text = oracle.synthtext(20 << 20)
    zeros = b"\0" * (24 << 20) + b"end"
    for level in (9, 1, 9, 2, 5, 1):
        assert bzx.compress_buffer(text, level) == bz2.compress(text, level), level
        assert bzx.compress_buffer(zeros, level) == bz2.compress(zeros, level), level


def test_concatenated_streams(bzx, oracle, tmp_path):
    """A .bz2 made of several streams (pbzip2 output, cat a.bz2 b.bz2): bzx_decompress_buffer decodes every one of
    them (the reference stops at the first footer, decompress.rs:81-95; bzip2 itself does not); bytes behind the
    footer that are no stream are ignored, as bzip2 does; and the command line tool never deletes the source after a
    failed decode."""
    import subprocess
    from bzx_ctypes import ROOT, BzxError
    a, b, c = oracle.synthtext(1_200_000), b"\0" * 70_000 + oracle.randbytes(30_001), b""
    z = bz2.compress(a, 9) + bz2.compress(b, 1) + bz2.compress(c, 5) + bz2.compress(a[:777], 3)
    assert bz2.decompress(z) == a + b + c + a[:777]
    assert bzx.decompress_buffer(z) == a + b + c + a[:777]
    assert bzx.decompress_buffer(z, cap=1 << 16) == a + b + c + a[:777]          # (grows through BZX_E_OUTBUF)
    assert bzx.decompress_buffer(bz2.compress(a, 9) + b"\0\0trailing bytes") == a
    one = bz2.compress(b, 9)
    bad = bytearray(one + one)
    bad[len(one) + len(one) // 2] ^= 0x10
    with pytest.raises(BzxError):
        bzx.decompress_buffer(bytes(bad))
    exe = os.path.join(ROOT, "bzip2-rust_amd", "bzx")
    p = tmp_path / "cat.bz2"
    p.write_bytes(z)
    subprocess.check_call([exe, "-d", str(p)])
    assert (tmp_path / "cat").read_bytes() == a + b + c + a[:777] and not p.exists()
    p.write_bytes(bytes(bad))
    assert subprocess.run([exe, "-d", "-f", str(p)], stderr=subprocess.PIPE).returncode != 0
    assert p.exists() and p.read_bytes() == bytes(bad)                           # the source survives a failed decode


def test_chunked_split(bzx, oracle):
    """bzx_split_rle1_chunk == one-shot split, for pieces of any size."""
    runs = (b"\xff" * 1000 + b"abc") * 2500 + b"\0" * 1_000_000 + oracle.synthtext(1_900_000)
    want = oracle.split_rle1(runs, 9)
    for chunk in (65536, 899981, 1 << 20, 5_000_000):
        assert bzx.split_rle1_chunks(runs, 9, chunk) == want, chunk


def test_decompress_libbz2_streams_and_own(bzx, oracle):
    """Device decompression (decompress.rs:38-404): streams made by libbz2 and by the device itself
    decode to the input; block CRCs and the combined CRC are checked on the way; damaged streams are refused."""
    from bzx_ctypes import BzxError
    from gen_golden import make_input
    for name, blk in _cases(oracle):
        for level in (1, 9):
            assert bzx.decompress_buffer(bz2.compress(blk, level), cap=len(blk) + 16) == blk, (name, level)
    for name in ("empty", "config1_text_1MiB_l1", "random_3MiB_l9", "zeros_2MiB_l9", "runs_mixed_l9", "allbytes_l1"):
        g = GOLDEN["streams"][name]
        data = make_input(oracle, g["input"])
        assert bzx.decompress_buffer(bz2.compress(data, g["level"]), cap=len(data) + 16) == data, name
    # the device's own output, hundreds of blocks; and the size is reported when the buffer is too small
    big = oracle.synthtext(200 << 20)
    z = bzx.compress_buffer(big, 9)
    assert bzx.decompress_buffer(z, cap=1 << 20) == big
    zeros = b"\0" * (100 << 20) + b"x"
    assert bzx.decompress_buffer(bzx.compress_buffer(zeros, 9), cap=len(zeros)) == zeros
    z = bytearray(bz2.compress(oracle.synthtext(3_000_000), 9))
    for damage in (len(z) // 2, 12, len(z) - 3):
        bad = bytearray(z)
        bad[damage] ^= 0x04
        with pytest.raises(BzxError):
            bzx.decompress_buffer(bytes(bad))
    with pytest.raises(BzxError):
        bzx.decompress_buffer(bytes(z[:-9]))
    with pytest.raises(BzxError):
        bzx.decompress_buffer(b"not a bzip2 stream at all")


def test_command_line_tool(oracle, tmp_path):
    """bzx (tools/bzx.cpp; reference src/tools/cli.rs:113-303, main.rs:30-34): -z/-d/-t, levels, -c, -k, -f, stdin/stdout;
    the files it writes are libbz2's, byte for byte."""
    import subprocess
    from bzx_ctypes import ROOT
    exe = os.path.join(ROOT, "bzip2-rust_amd", "bzx")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    data = oracle.synthtext(2_500_000) + b"\0" * 100_000
    p = tmp_path / "a.txt"
    p.write_bytes(data)
    subprocess.check_call([exe, "-k", "-5", str(p)])
    assert (tmp_path / "a.txt.bz2").read_bytes() == bz2.compress(data, 5) and p.exists()
    assert subprocess.run([exe, str(p)]).returncode != 0                        # output exists, no -f
    subprocess.check_call([exe, "-f", "--best", str(p)])
    assert (tmp_path / "a.txt.bz2").read_bytes() == bz2.compress(data, 9) and not p.exists()
    subprocess.check_call([exe, "-t", str(tmp_path / "a.txt.bz2")])
    subprocess.check_call([exe, "-d", str(tmp_path / "a.txt.bz2")])
    assert p.read_bytes() == data and not (tmp_path / "a.txt.bz2").exists()
    z = subprocess.run([exe, "-c", "-1"], input=data, stdout=subprocess.PIPE, check=True).stdout
    assert z == bz2.compress(data, 1)
    assert subprocess.run([exe, "-dc"], input=z, stdout=subprocess.PIPE, check=True).stdout == data
    bad = bytearray(z)
    bad[len(bad) // 3] ^= 1
    assert subprocess.run([exe, "-t"], input=bytes(bad), stderr=subprocess.PIPE).returncode != 0


def test_large_roundtrip_properties(bzx, oracle):
    """Size-independent properties at a multi-hundred-block size (BASELINE configs[2] shape, scaled to keep the
    test short): libbz2 decodes the device stream back to the input; block count and framing are right; the first
    blocks are byte-identical to the oracle's stream of a prefix."""
    n = 272 << 20       # 317 blocks > 256 compute units: the sort's partial last round runs beside the MTF stage
    data = oracle.synthtext(n)
    out = bzx.compress_buffer(data, 9)
    st = bzx.stats()
    assert out[:4] == b"BZh9" and st.nblk == (n + 899980) // 899981
    assert bz2.decompress(out) == data
    # prefix property: all blocks but the last of a prefix stream appear unchanged (bit-exact) at the start
    pre = 8 * 899981 + 12345
    ref, nb = oracle.compress(data[:pre], 9)
    assert nb == 9
    # the first 8 blocks end on a bit boundary; compare whole bytes safely inside them
    assert out[: len(ref) - 300_000] == ref[: len(ref) - 300_000]


def test_block_info(bzx, oracle):
    """bzx_get_block_info: the per-block figures the reference logs at -vvv (compress_block.rs:58-63,
    huffman.rs:176-181), against the oracle's stages block by block."""
    data = oracle.synthtext(250_000) + b"\0" * 3000 + bytes(range(256)) * 40
    out = bzx.compress_buffer(data, 1)
    assert out == bz2.compress(data, 1)
    blocks = oracle.split_rle1(data, 1)
    assert bzx.stats().nblk == len(blocks)
    total = 0
    for i, (img, crc) in enumerate(blocks):
        bi = bzx.block_info(i)
        L, orig = oracle.bwt(img)
        mo, fo, iuo, niu = oracle.mtf(L)
        hf = oracle.huff(mo, fo, niu + 2)
        assert (bi.n, bi.crc, bi.orig_ptr, bi.n_in_use, bi.n_mtf) == (len(img), crc, orig, niu, len(mo)), i
        assert bi.n_tables == hf[0] and bi.n_selectors == len(hf[1]), i
        assert bi.bits == 48 + 32 + 1 + 24 + bi.bits_symbol_map + 3 + 15 + bi.bits_selectors + bi.bits_tables + bi.bits_payload
        total += bi.bits
    assert (32 + total + 80 + 7) // 8 == len(out)


def test_leftover_groups_paths(bzx, oracle):
    """The three ways a block's deep repeats are finished, each checked to be the one taken: (a) duplicated files in
    text -- buckets give up, the grid-wide rank rounds close all of them; (b) thousands of rotations sharing a long
    prefix -- an oversized bin is left as one group and the general sorter finishes the block; (c) a block that is a
    whole number of copies of a unit -- refused by the split kernel, sorted from scratch, flagged periodic."""
    big = oracle.synthtext(900_000)
    dup = big[:400_000] + big[100_000:180_000] + big[400_000:700_000] + big[120_000:150_000] + b"!"
    out = bzx.compress_buffer(dup, 9)
    st = bzx.stats()
    assert out == bz2.compress(dup, 9)
    assert st.n_open_buckets > 0 and st.n_open_left == 0 and st.n_resume_left == 0 and st.n_from_scratch == 0
    assert st.n_unsorted == 0       # the optimistic initial sort's check held for every bucket (wg_radix_sort_opt)
    runs = big[:120_000] + (b"ab" * 700 + b"c") * 200 + big[120_000:200_000] + b"?"
    out = bzx.compress_buffer(runs, 9)
    st = bzx.stats()
    assert out == bz2.compress(runs, 9)
    assert st.n_resume_left == 1 and st.n_from_scratch == 0
    per = oracle.synthtext(90_000) * 5
    out = bzx.compress_buffer(per, 9)
    st = bzx.stats()
    assert out == bz2.compress(per, 9)
    assert st.n_periodic == 1 and st.n_resume_left == 1 and st.n_from_scratch == 0
    tiny = b"abcabcabd" * 5
    out = bzx.compress_buffer(tiny, 9)
    st = bzx.stats()
    assert out == bz2.compress(tiny, 9)
    assert st.n_periodic == 1 and st.n_resume_left == 1
    # (d) thousands of copies of one string with different continuations: oversized groups that the regrouping pass
    # deals into rank-round items -- nothing left to the general sorter, nothing refused (alone and inside text)
    rnd = random.Random(5)
    unit = rnd.randbytes(65)
    copies = b"".join(unit + rnd.randbytes(4) for _ in range(2100))
    for data in (copies, big[:300_000] + copies + big[300_000:600_000]):
        out = bzx.compress_buffer(data, 9)
        st = bzx.stats()
        assert out == bz2.compress(data, 9)
        assert st.n_open_buckets > 0 and st.n_resume_left == 0 and st.n_from_scratch == 0


def test_deep_repeats(bzx, oracle):
    """Highly redundant blocks (near-identical copies, fixed-size records): large groups that single symbols do
    not separate are frozen and finished by prefix doubling; the stream must still be libbz2's bit for bit."""
    import random
    rnd = random.Random(5)
    base = bytearray(oracle.synthtext(64000))
    copies = bytearray()
    for _ in range(14):                                  # 14 copies with ~0.5 % of the bytes changed
        c = bytearray(base)
        for _ in range(300):
            c[rnd.randrange(len(c))] = rnd.randrange(256)
        copies += c
    rec = rnd.randbytes(190)
    records = b"".join(rec + b"%010d" % (i * 7919) for i in range(4000))
    twice = oracle.synthtext(440000) * 2 + b"tail"        # every rotation has a twin 440,000 bytes away
    # 40 four-byte prefixes x 300 occurrences with (almost) distinct next bytes: oversized groups that the symbol
    # splitter turns into singletons, i.e. sort tiles with more than 128 groups (8-bit tile-local group index)
    parts = [bytes([65 + p, 66 + p, 67 + p, 68 + p, i & 255, (i * 37 + p) & 255]) + rnd.randbytes(5)
             for p in range(40) for i in range(300)]
    rnd.shuffle(parts)
    singles = b"".join(parts) + bytes(range(256))
    for data in (bytes(copies), records, twice, bytes(copies) + records[:200000], singles):
        assert bzx.compress_buffer(data, 9) == bz2.compress(data, 9)


def test_real_files_from_the_image(bzx):
    """Real data (python sources: 7-bit text with deep repeats; shared objects: 8-bit, highly redundant fat binaries)
    read from the system image -- the kind of input that synthetic generators miss."""
    import glob

    def collect(patterns, limit):
        out, seen = bytearray(), set()
        for pat in patterns:
            for f in sorted(glob.glob(pat, recursive=True)):
                rp = os.path.realpath(f)
                if rp in seen or not os.path.isfile(rp):
                    continue
                seen.add(rp)
                try:
                    with open(rp, "rb") as fh:
                        out += fh.read(limit - len(out))       # some of these libraries are gigabytes long
                except OSError:
                    continue
                if len(out) >= limit:
                    return bytes(out[:limit])
        return bytes(out)

    sets = [collect(["/usr/lib/python3*/**/*.py"], 6 << 20),
            collect(["/opt/rocm/lib/*.so*", "/usr/lib/x86_64-linux-gnu/*.so*"], 16 << 20)]
    if not any(len(d) > (1 << 20) for d in sets):
        pytest.skip("no system files to read")
    for data in sets:
        if len(data) > (1 << 20):
            assert bzx.compress_buffer(data, 9) == bz2.compress(data, 9)


def test_structured_fuzz_slice(bzx):
    """A fixed slice of the differential fuzzer (tests/gpu_probe_fuzz.py: alphabets, repeats at several scales,
    records, runs, near-periodic data, twins, mixtures) against libbz2."""
    from gpu_probe_fuzz import gen
    for c in range(120):
        rnd = random.Random(7 * 100003 + c)
        data = gen(rnd)
        level = rnd.choice([1, 1, 2, 9])
        assert bzx.compress_buffer(data, level) == bz2.compress(data, level), c


def test_block_boundary_fuzz_slice(bzx):
    """A fixed slice of tests/gpu_probe_fuzz_big.py: 0.1-4 MB inputs at all levels, long runs across block
    boundaries, inputs that end at or just past a full block."""
    from gpu_probe_fuzz_big import big
    for c in range(40):
        rnd = random.Random(5 * 7919 + c)
        data = big(rnd)
        level = rnd.randrange(1, 10)
        assert bzx.compress_buffer(data, level) == bz2.compress(data, level), c


def test_alphabet_sizes(bzx, oracle):
    """Every symbol width of the packed block (1..8 bits) and both sides of each power of two, on skewed and on
    repetitive data (level 1: several blocks per input)."""
    import random
    rnd = random.Random(11)
    for k in (1, 2, 3, 4, 5, 8, 9, 16, 17, 32, 33, 64, 65, 128, 129, 200, 256):
        syms = rnd.sample(range(256), k)
        weights = [1.0 / (i + 1) for i in range(k)]
        body = bytes(rnd.choices(syms, weights, k=120000))
        data = body + body[:40000] + bytes(rnd.choices(syms, k=60000)) + body[5000:65000]
        assert bzx.compress_buffer(data, 1) == bz2.compress(data, 1), k


def test_random_and_zero_heavy_inputs(bzx, oracle):
    """BASELINE configs[4] shapes at reduced size: incompressible bytes (sort worst case for MTF) and an
    all-zero input (RLE1-heavy, periodic blocks)."""
    rnd = oracle.randbytes(24 << 20)
    out = bzx.compress_buffer(rnd, 9)
    assert out == bz2.compress(rnd, 9)
    z = b"\0" * (48 << 20)          # 1 periodic block of 899,985 bytes + tail
    out = bzx.compress_buffer(z, 9)
    assert out == bz2.compress(z, 9)


def test_shard_entry_points_single_rank(bzx, oracle):
    """bzx_shard_prepare / bzx_shard_emit_packed / bzx_shard_assemble_*  with world size 1 on the
    device: same stream as libbz2.
    (World size 2 is covered on CPU over gloo in tests/test_shard_gloo.py.)"""
    import ctypes as C
    import torch
    L = bzx.lib
    L.bzx_shard_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_uint32,
                                    C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t]
    L.bzx_shard_emit_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_size_t)]
    L.bzx_shard_assemble_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.bzx_shard_assemble_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    data = oracle.synthtext(5_000_000) + b"\0" * 70000 + oracle.randbytes(300000)
    d_raw = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    bits = torch.zeros(64, dtype=torch.int64, device="cuda")
    cap = len(data) + len(data) // 4 + 65536
    d_out = torch.full((cap,), 255, dtype=torch.uint8, device="cuda")
    d_packed = torch.full((cap,), 255, dtype=torch.uint8, device="cuda")
    nblk = C.c_uint32()
    torch.cuda.synchronize()
    bzx._check(L.bzx_shard_prepare(bzx.ctx, d_raw.data_ptr(), len(data), 9, 0, 1, C.byref(nblk), bits.data_ptr(), 64))
    pl, ol = C.c_size_t(), C.c_size_t()
    bzx._check(L.bzx_shard_emit_packed(bzx.ctx, bits.data_ptr(), d_packed.data_ptr(), cap, C.byref(pl), C.byref(ol)))
    bzx._check(L.bzx_shard_assemble_begin(bzx.ctx, d_out.data_ptr(), cap, None))
    bzx._check(L.bzx_shard_assemble_rank(bzx.ctx, d_packed.data_ptr(), 0, d_out.data_ptr()))
    bzx._check(L.bzx_ctx_sync(bzx.ctx))
    torch.cuda.synchronize()
    out = d_out[:ol.value].cpu().numpy().tobytes()
    assert nblk.value == 6 and out == bz2.compress(data, 9)
    # the same with the sharded split analysis (bzx_shard_scan_* + bzx_shard_prepare_scanned) on
    # run-heavy input -- runs of 1..600, a run across several blocks' worth of input -- at world 1 (the tile arrays of
    # "all ranks" are this rank's; worlds 2 and 4, with runs across the ranks' borders: tests/test_shard_gloo.py)
    from gen_golden import make_input
    L.bzx_shard_scan_entries.restype = C.c_size_t
    L.bzx_shard_scan_entries.argtypes = [C.c_size_t, C.c_uint32]
    for f in (L.bzx_shard_scan_runs, L.bzx_shard_scan_counts):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p]
    L.bzx_shard_prepare_scanned.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p,
                                            C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t]
    data = (b"\0" * 3_000_000 + make_input(oracle, {"kind": "runs", "n": 4_000_000, "seed": 9}) + oracle.synthtext(2_000_000) +
            b"\xee" * 700_000 + oracle.randbytes(100_000))
    for level in (9, 2):
        want = bz2.compress(data, level)
        d_raw = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        P = L.bzx_shard_scan_entries(len(data), 1)
        tiles = torch.full((3, 1, P), -5, dtype=torch.int64, device="cuda")
        bits = torch.zeros(256, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        bzx._check(L.bzx_shard_scan_runs(bzx.ctx, d_raw.data_ptr(), len(data), 0, 1, tiles.data_ptr()))
        bzx._check(L.bzx_shard_scan_counts(bzx.ctx, d_raw.data_ptr(), len(data), 0, 1, tiles.data_ptr()))
        bzx._check(L.bzx_shard_prepare_scanned(bzx.ctx, d_raw.data_ptr(), len(data), level, 0, 1, tiles.data_ptr(),
                                               C.byref(nblk), bits.data_ptr(), 256))
        bzx._check(L.bzx_shard_emit_packed(bzx.ctx, bits.data_ptr(), d_packed.data_ptr(), cap, C.byref(pl), C.byref(ol)))
        bzx._check(L.bzx_shard_assemble_begin(bzx.ctx, d_out.data_ptr(), cap, None))
        bzx._check(L.bzx_shard_assemble_rank(bzx.ctx, d_packed.data_ptr(), 0, d_out.data_ptr()))
        bzx._check(L.bzx_ctx_sync(bzx.ctx))
        torch.cuda.synchronize()
        assert d_out[:ol.value].cpu().numpy().tobytes() == want, level
        assert nblk.value == len(oracle.split_rle1(data, level))


def test_output_buffer_too_small(bzx, oracle):
    import ctypes as C
    from bzx_ctypes import BzxError
    data = oracle.randbytes(200000)
    out = C.create_string_buffer(1000)
    ol = C.c_size_t()
    rc = bzx.lib.bzx_compress_buffer(bzx.ctx, data, len(data), 9, out, 1000, C.byref(ol))
    assert rc == -4 and ol.value > 200000           # BZX_E_OUTBUF, needed size reported
    blk = oracle.synthtext(50000)
    ob = C.create_string_buffer(100)
    olen = C.c_size_t()
    pad = C.c_uint8()
    rc = bzx.lib.bzx_compress_block(bzx.ctx, blk, len(blk), oracle.crc32(blk), ob, 100, C.byref(olen), C.byref(pad))
    assert rc == -4
