"""CPU tests (-m "not gpu"): kernel LOGIC through the fiber emulator (tests/emu).

The HIP sources in bzip2-rust_amd/csrc are compiled with g++ against tests/emu/hip/hip_runtime.h (workgroups as
fibers, wave64 collectives, LDS as static storage) and driven through the same C ABI as on the GPU.  This proves
indexing, scans and loop bounds of the kernels on small inputs in the GPU-less build container; it proves
nothing about the device (memory ordering, performance) -- the -m gpu suite does that."""
import bz2
import os
import random
import subprocess

import pytest

from bzx_ctypes import EMU_PATH, ROOT, BzxLib


@pytest.fixture(scope="module")
def emu():
    csrc = os.path.join(ROOT, "bzip2-rust_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(ROOT, "tests", "emu", "hip", "hip_runtime.h")]
    if not os.path.exists(EMU_PATH) or any(os.path.getmtime(s) > os.path.getmtime(EMU_PATH) for s in srcs):
        subprocess.check_call(["bash", os.path.join(ROOT, "tests", "emu", "build_emu.sh")])
    lib = BzxLib(EMU_PATH)
    yield lib
    lib.close()


def test_emu_stages_small(emu, oracle):
    rnd = random.Random(4)
    cases = [b"a", b"aa", b"banana", b"Making a silly test.", oracle.synthtext(3000), rnd.randbytes(2500),
             bytes(rnd.choice(b"ab") for _ in range(3000)), b"abcabcd" * 15, b"\0\0\0\0\xfb" * 300 + b"\0\0\0\0\x07",
             oracle.synthtext(1500) * 2 + b"z"]
    for blk in cases:
        L, orig, _ = emu.stage_bwt(blk)
        assert (L, orig) == oracle.bwt(blk)
        mo, fo, iuo, niu = oracle.mtf(L)
        assert emu.stage_mtf(L) == (mo, fo, iuo)
        assert emu.stage_huffman(mo, fo, niu + 2) == oracle.huff(mo, fo, niu + 2)
        crc = oracle.crc32(blk)
        assert emu.compress_block(blk, crc) == oracle.compress_block(blk, crc)


def test_emu_alphabet_sizes(emu, oracle):
    """Symbol widths 1..8 bits of the packed block the sorter builds."""
    rnd = random.Random(21)
    for k in (1, 2, 3, 5, 9, 17, 33, 65, 129, 256):
        syms = rnd.sample(range(256), k)
        body = bytes(rnd.choices(syms, [1.0 / (i + 1) for i in range(k)], k=1800))
        blk = body + body[:700] + bytes(rnd.choices(syms, k=500))
        L, orig, _ = emu.stage_bwt(blk)
        assert (L, orig) == oracle.bwt(blk), k


def test_emu_many_groups_per_tile(emu, oracle):
    """Oversized groups that the symbol splitter turns into singletons: sort tiles with more than 128 groups."""
    rnd = random.Random(3)
    parts = [bytes([65 + p, 66 + p, 67 + p, 68 + p, i & 255, (i * 37 + p) & 255]) + rnd.randbytes(5)
             for p in range(40) for i in range(300)]
    rnd.shuffle(parts)
    blk = b"".join(parts) + bytes(range(256))
    L, orig, _ = emu.stage_bwt(blk)
    assert (L, orig) == oracle.bwt(blk)


def test_emu_block_info(emu, oracle):
    """bzx_get_block_info: the per-block figures the reference logs at -vvv (compress_block.rs:58-63,
    huffman.rs:176-181), against the oracle's stages block by block."""
    data = oracle.synthtext(120_000) + b"\0" * 3000 + bytes(range(256)) * 40
    out = emu.compress_buffer(data, 1)
    assert out == bz2.compress(data, 1)
    blocks = oracle.split_rle1(data, 1)
    assert emu.stats().nblk == len(blocks)
    total = 0
    for i, (img, crc) in enumerate(blocks):
        bi = emu.block_info(i)
        L, orig = oracle.bwt(img)
        mo, fo, iuo, niu = oracle.mtf(L)
        hf = oracle.huff(mo, fo, niu + 2)
        assert (bi.n, bi.crc, bi.orig_ptr, bi.n_in_use, bi.n_mtf) == (len(img), crc, orig, niu, len(mo)), i
        assert bi.n_tables == hf[0] and bi.n_selectors == len(hf[1]), i
        assert bi.bits == 48 + 32 + 1 + 24 + bi.bits_symbol_map + 3 + 15 + bi.bits_selectors + bi.bits_tables + bi.bits_payload
        total += bi.bits
    assert (32 + total + 80 + 7) // 8 == len(out)


def test_emu_short_rank_rounds(emu, oracle):
    """Buckets that are still open after the rank rounds although the rounds resolved (and moved) some of their ranks:
    the general sorter finishes the block and every row of the last column must be current.  Built with two rank
    rounds instead of fourteen (tests/emu/build_emu.sh) so that repeats of a few hundred symbols get there: three
    copies of a 600-symbol stretch stay tied, three copies of its first 120 symbols -- different bytes before and after
    each -- are separated from them in the rounds."""
    lib = BzxLib(EMU_PATH.replace("libbzx_emu.so", "libbzx_emu_rk2.so"))
    try:
        for seed in range(3):
            rnd = random.Random(seed)
            d = oracle.synthtext(4000)
            m, s = d[300:900], d[300:420]
            parts = [d]
            for i in range(3):
                parts += [b"QZA"[i:i + 1], m, bytes([rnd.randrange(1, 30)])]
            for i in range(3):
                parts += [b"KBX"[i:i + 1], s, bytes([rnd.randrange(1, 30)]), d[rnd.randrange(1000, 3000):][:50]]
            blk = b"".join(parts)
            assert lib.compress_buffer(blk, 9) == bz2.compress(blk, 9), seed
            st = lib.stats()
            assert st.n_open_left > 0 and st.n_resume_left == 1 and st.n_periodic == 0
    finally:
        lib.close()


def test_emu_leftover_groups_paths(emu, oracle):
    """Deep repeats: buckets that give up and are closed by the rank rounds; an oversized bin left as one group for the
    general sorter; a periodic block sorted from scratch (the paths are asserted, as in the device test)."""
    big = oracle.synthtext(30000)
    dup = big[:15000] + big[2000:9000] + big[15000:] + big[2000:5000] + b"!"
    assert emu.compress_buffer(dup, 9) == bz2.compress(dup, 9)
    st = emu.stats()
    assert st.n_open_buckets > 0 and st.n_open_left == 0 and st.n_resume_left == 0 and st.n_from_scratch == 0
    assert st.n_unsorted == 0          # the optimistic initial sort's check never failed (bzx_bsort.hip, wg_radix_sort_opt)
    runs = (b"ab" * 700 + b"c") * 6 + b"d"

    assert emu.compress_buffer(runs, 9) == bz2.compress(runs, 9)
    st = emu.stats()
    assert st.n_resume_left == 1 and st.n_from_scratch == 0
    per = big[:2500] * 3
    assert emu.compress_buffer(per, 9) == bz2.compress(per, 9)
    st = emu.stats()
    assert st.n_periodic == 1


def test_emu_oversized_group_regrouped(emu):
    """2,100 copies of one 65-byte string, each followed by four random bytes: the rotations at the start of a copy
    agree on more than the split's whole depth with 2,099 others -- a group too big for a workgroup's rank round.  The
    regrouping pass (bzx_brank_giant_kernel) deals it into ordinary rank-round items by the ranks ahead, and the rounds
    finish the block: nothing is left to the general sorter, and the block is not refused although the split emits
    thousands of one-bucket splits for it (a lone block's share of the work lists)."""
    rnd = random.Random(5)
    unit = rnd.randbytes(65)
    data = b"".join(unit + rnd.randbytes(4) for _ in range(2100))
    assert emu.compress_buffer(data, 9) == bz2.compress(data, 9)
    st = emu.stats()
    assert st.n_open_buckets > 0 and st.n_open_left == 0 and st.n_resume_left == 0 and st.n_from_scratch == 0


def test_emu_concurrent_compress_block(emu, oracle):
    """bzx_compress_block from several host threads on one context (the reference calls compress_block from every
    rayon worker, compress.rs:125-132): calls are collected into device batches, every caller gets its own result."""
    import threading
    rnd = random.Random(17)
    blocks = [oracle.synthtext(900 + 137 * i) if i % 3 else rnd.randbytes(700 + 91 * i) for i in range(12)]
    want = [oracle.compress_block(b, oracle.crc32(b)) for b in blocks]
    got = [None] * len(blocks)

    def work(i):
        got[i] = emu.compress_block(blocks[i], oracle.crc32(blocks[i]))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(blocks))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert got == want


def test_emu_split_and_stream(emu, oracle):
    rnd = random.Random(8)
    runs = bytearray()
    while len(runs) < 150000:
        runs += bytes([rnd.choice(b"ab\0")]) * rnd.randint(1, 400)
    big = oracle.synthtext(520000)          # several plain blocks: the batched fast path of the boundary chain
    mixed = big[:260000] + b"\0" * 5000 + big[260000:] + b"ab" * 3      # ... and a long run that interrupts it
    for data, level in ((oracle.synthtext(130000), 1), (bytes(runs), 1), (b"", 9), (b"xyz", 9), (big, 1), (mixed, 1)):
        assert emu.split_rle1(data, level) == oracle.split_rle1(data, level)
    data = oracle.synthtext(60000) + b"\0" * 2000 + oracle.synthtext(45000)
    assert emu.compress_buffer(data, 1) == bz2.compress(data, 1)


def test_emu_chunked_stream_and_chunked_split(emu, oracle):
    """bzx_cstream_* / bzx_split_rle1_chunk (RLE1Block over a source that arrives in pieces, rle1.rs:49-85,245-263):
    chunk borders inside long runs, at block-full edges, one-byte chunks, an empty final call."""
    rnd = random.Random(3)
    runs = bytearray()
    while len(runs) < 220000:
        runs += bytes([rnd.choice(b"ab\0")]) * rnd.randint(1, 700)
    # (sizes: just over two level-1 blocks; the emulator runs ~10 us per lane-step)
    cases = [(b"", 1, 1000), (b"xyz", 9, 2), (oracle.synthtext(205000), 1, 70000), (bytes(runs), 1, 33333),
             (b"\0" * 700000, 1, 300000),
             (oracle.synthtext(99981) + b"\0" * 5000 + oracle.synthtext(20000), 1, [1, 99980, 4999, 3, 50000])]
    for data, level, chunk in cases:
        assert emu.cstream_compress(data, level, chunk) == bz2.compress(data, level), (len(data), chunk)
    for data, level, chunk in cases[1:5]:
        assert emu.split_rle1_chunks(data, level, chunk) == oracle.split_rle1(data, level), (len(data), chunk)


def test_emu_decompress(emu, oracle):
    """bzx_decompress_buffer (decompress.rs:38-404): libbz2-made streams come back as the input; damage is reported."""
    from bzx_ctypes import BzxError
    rnd = random.Random(9)
    # (the emulator spends ~0.6 ms per compressed byte here: the two-block case is a two-letter text)
    cases = [(b"", 9), (b"xyz", 9), (b"Making a silly test.", 9), (oracle.synthtext(12000), 1), (rnd.randbytes(6000), 1),
             (b"\0" * 30000 + b"ab" * 300 + b"\xff" * 1000, 1), (bytes(rnd.choice(b"ab") for _ in range(101000)), 1),
             (bytes(rnd.choice(b"abc") for _ in range(9000)), 9)]
    for data, level in cases:
        assert emu.decompress_buffer(bz2.compress(data, level)) == data, (len(data), level)
    z = bytearray(bz2.compress(oracle.synthtext(8000), 1))
    z[len(z) // 2] ^= 0x10
    with pytest.raises(BzxError):
        emu.decompress_buffer(bytes(z))
    with pytest.raises(BzxError):
        emu.decompress_buffer(bz2.compress(oracle.synthtext(8000), 1)[:-7])
