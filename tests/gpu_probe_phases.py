"""Ad-hoc GPU probe: BWT kernel phase breakdown (diagnostic timers) for a batch of identical blocks."""
import sys, ctypes as C
sys.path.insert(0, "tests")
from bzx_ctypes import *
o = Oracle(); lib = BzxLib()
L = lib.lib
L.bzx_dbg_time_stages.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_int, C.POINTER(C.c_float)]
L.bzx_dbg_phase_timers.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1194
if kind.startswith("file:"):
    blk = open(kind[5:], "rb").read()
elif kind == "py":
    import glob
    buf = bytearray()
    for f in sorted(glob.glob("/usr/lib/python3*/**/*.py", recursive=True)):
        buf += open(f, "rb").read()
        if len(buf) > 3_000_000: break
    blk = o.split_rle1(bytes(buf[900000:2300000]), 9)[0][0]
else:
    blk = o.synthtext(899981) if kind == "text" else o.randbytes(899981)
ms = (C.c_float * 4)(); t = (C.c_ulonglong * 64)()
lib._check(L.bzx_dbg_time_stages(lib.ctx, blk, len(blk), reps, 1, ms))
lib._check(L.bzx_dbg_phase_timers(lib.ctx, 1, None))
lib._check(L.bzx_dbg_time_stages(lib.ctx, blk, len(blk), reps, 15, ms))
lib._check(L.bzx_dbg_phase_timers(lib.ctx, 0, t))
names = {4: "RANK re-rank (ISA scatter)", 6: "RANK tile sorts (ISA gather)", 7: "RANK big-group split (key2 digits)", 5: "ISA build", 48: "EMIT load tables + header", 49: "EMIT selectors", 50: "EMIT coding tables", 51: "EMIT payload", 40: "HUF init tables", 41: "HUF cost+rfreq passes (x4)", 42: "HUF code lengths (x4)", 43: "HUF codes+store", 44: "HUF payload sizes", 45: "HUF selector MTF", 32: "MTF in-use", 33: "MTF recency lists", 34: "MTF start lists", 35: "MTF ranks", 36: "MTF zero-run + emit", 0: "I1 build", 1: "I2 4 radix passes", 2: "R0 rerank", 3: "F final gather"}
for r in range(8):
    names[8 + 3 * r] = f"round{r} seg sort"; names[9 + 3 * r] = f"round{r} rerank"; names[10 + 3 * r] = f"round{r} big-group split"
cnt = list(t[52:64])
for i in range(52, 64):
    t[i] = 0
tot = sum(t)
print(f"bwt {ms[0]:.2f} mtf {ms[1]:.2f} huf {ms[2]:.2f} emit {ms[3]:.2f} ms for {reps} blocks; phase ticks summed over blocks (per block ms = ticks/100e3/reps):")
for i in range(64):
    if t[i]:
        print(f"  {names.get(i, i):28s} {t[i] / 100e3 / reps:8.3f} ms/block  {100.0 * t[i] / tot:5.1f}%")
print(f"  total {tot / 100e3 / reps:.3f} ms/block")
for r in range(6):
    if cnt[2 * r]:
        print(f"  RANK round {r}{'+' if r == 5 else ' '}: unresolved {cnt[2 * r] / reps:10.0f}  largest group {cnt[2 * r + 1] / reps:9.0f}  (per block; round 5+ summed)")
