"""Ad-hoc GPU probe (not a pytest): per-stage and per-block parity + stage timings on the device."""
import sys, time, ctypes as C
sys.path.insert(0, "tests")
from bzx_ctypes import *
o = Oracle()
lib = BzxLib()
lib.lib.bzx_dbg_time_stages.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_int, C.POINTER(C.c_float)]
def check(name, c):
    Lo, oo = o.bwt(c)
    L, orig, st = lib.stage_bwt(c)
    ok0 = (L == Lo) and (orig == oo or st == 1)
    mt, fr, iu = lib.stage_mtf(Lo)
    mo, fo, iuo, niu = o.mtf(Lo)
    ok1 = (mt == mo and fr == fo and iu == iuo)
    ng, sel, lens, codes = lib.stage_huffman(mo, fo, niu + 2)
    ngo, selo, lenso, codeso = o.huff(mo, fo, niu + 2)
    ok2 = (ng == ngo and sel == selo and lens == lenso and codes == codeso)
    crc = o.crc32(c)
    t = time.time(); img, pad = lib.compress_block(c, crc); dt = time.time() - t
    imgo, pado = o.compress_block(c, crc)
    ok3 = (img == imgo and pad == pado)
    st_ = lib.stats()
    print(f"{name:14s} n={len(c):7d} bwt={ok0} mtf={ok1} huff={ok2} block={ok3} periodic={st} img={len(img)} "
          f"ms bwt/mtf/huf/emit = {st_.ms_bwt:.2f}/{st_.ms_mtf:.2f}/{st_.ms_huffman:.2f}/{st_.ms_emit:.2f} wall={dt*1e3:.1f}", flush=True)
    if not ok1:
        print("   nmtf", len(mt), len(mo), [(i, a, b) for i, (a, b) in enumerate(zip(mt, mo)) if a != b][:5])
    if not ok3:
        print("   first diffs", [i for i, (a, b) in enumerate(zip(img, imgo)) if a != b][:5], len(img), len(imgo), pad, pado)
    return ok0 and ok1 and ok2 and ok3
import random
random.seed(5)
ok = True
ok &= check("tiny", b"banana")
ok &= check("silly", b"Making a silly test.")
ok &= check("text20k", o.synthtext(20000))
ok &= check("lowent", bytes(random.choice(b"ab") for _ in range(60000)))
ok &= check("all256", random.randbytes(100000))
text = o.synthtext(899981)
ok &= check("text900k", text)
rnd = o.randbytes(899981)
ok &= check("rand900k", rnd)
ok &= check("zeromix", b"\0" * 300000 + o.synthtext(300000) + b"\xff" * 299000)
print("PARITY", "OK" if ok else "FAIL", flush=True)
ms = (C.c_float * 4)()
for name, blk in (("text", text), ("rand", rnd)):
    for reps in (1, 256, 1194):
        lib._check(lib.lib.bzx_dbg_time_stages(lib.ctx, blk, len(blk), reps, 15, ms))
        lib._check(lib.lib.bzx_dbg_time_stages(lib.ctx, blk, len(blk), reps, 15, ms))
        tot = sum(ms)
        print(f"{name} reps={reps:5d} bwt {ms[0]:8.2f} mtf {ms[1]:8.2f} huf {ms[2]:8.2f} emit {ms[3]:8.2f} ms -> {reps*len(blk)/tot/1e3:9.1f} MB/s", flush=True)
