"""Ad-hoc GPU probe (not a pytest): BWT stage parity + timing on the device."""
import sys, time, ctypes as C
sys.path.insert(0, "tests")
from bzx_ctypes import *
o = Oracle()
lib = BzxLib()
lib.lib.bzx_dbg_time_stages.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_int, C.POINTER(C.c_float)]
def check(name, blk):
    t = time.time(); L, orig, st = lib.stage_bwt(blk); dt = time.time() - t
    Lo, oo = o.bwt(blk)
    print(f"{name:20s} n={len(blk):7d} L_ok={L == Lo} orig={orig} oracle_orig={oo} status={st} wall={dt*1e3:.1f} ms", flush=True)
    return L == Lo
ok = True
ok &= check("tiny", b"banana")
ok &= check("text20k", o.synthtext(20000))
text = o.synthtext(899981)
ok &= check("text900k", text)
rnd = o.randbytes(899981)
ok &= check("rand900k", rnd)
ok &= check("zeros-like", (b"\0\0\0\0\xfb" * 179997))
ms = (C.c_float * 4)()
for name, blk in (("text", text), ("rand", rnd)):
    for reps in (1, 256, 512, 1194):
        lib._check(lib.lib.bzx_dbg_time_stages(lib.ctx, blk, len(blk), reps, 1, ms))
        lib._check(lib.lib.bzx_dbg_time_stages(lib.ctx, blk, len(blk), reps, 1, ms))
        print(f"{name} reps={reps:5d} bwt {ms[0]:9.2f} ms  -> {reps*len(blk)/ms[0]/1e3:9.1f} MB/s", flush=True)
print("PARITY", "OK" if ok else "FAIL")
