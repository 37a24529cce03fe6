"""Ad-hoc GPU probe: the near-periodic tail block of an all-zero input, a 64 MiB zero buffer end to end, and (BZX_LIB=...libbzx_diag.so)
the phase split of the periodic kernel; run it under `timeout`."""
import sys, time
sys.path.insert(0, "tests")
from bzx_ctypes import *
o = Oracle(); lib = BzxLib(max_blocks=16)
tail = b"\0\0\0\0\xfb" * 166350 + b"\0\0\0\0\x04"
for name, img in (("tail831k", tail), ("tail-small", b"\0\0\0\0\xfb" * 3000 + b"\0\0\0\0\x04")):
    print("start", name, len(img), flush=True)
    t = time.time(); L, orig, st = lib.stage_bwt(img); dt = time.time() - t
    Lo, oo = o.bwt(img)
    print(f"  {dt*1e3:.1f} ms  {'OK' if (L, orig) == (Lo, oo) else 'MISMATCH'} status {st}", flush=True)
import os
DIAG = "diag" in os.environ.get("BZX_LIB", "")
if DIAG:
    lib.lib.bzx_dbg_phase_timers.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib._check(lib.lib.bzx_dbg_phase_timers(lib.ctx, 1, None))
data = b"\0" * (64 << 20)
print("start 64MiB zeros", flush=True)
t = time.time(); z = lib.compress_buffer(data, 9); dt = time.time() - t
import bz2
print(f"  {dt*1e3:.1f} ms {'OK' if z == bz2.compress(data, 9) else 'MISMATCH'}", lib.stats().nblk, lib.stats().n_periodic, flush=True)
if DIAG:
    t = (C.c_ulonglong * 128)()
    lib._check(lib.lib.bzx_dbg_phase_timers(lib.ctx, 0, t))
    names = {110: "copy+setup", 111: "main_sort", 112: "fb-init", 113: "fb-eclass", 114: "fb-walk(lane0)", 115: "fb-allsame", 116: "fb-bigqsort", 117: "fb-total-rest"}
    print("  periodic kernel, ms summed over workgroups:", {v: round(t[k] / 1e5, 1) for k, v in names.items()})
