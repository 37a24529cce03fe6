"""Ad-hoc GPU probe: per-block time in the BWT kernel on real data; dumps the slowest block for offline analysis."""
import os, sys, glob, ctypes as C
sys.path.insert(0, "tests")
import numpy as np, torch
from bzx_ctypes import *
lib = BzxLib(max_blocks=400); L = lib.lib
L.bzx_dbg_phase_timers.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
L.bzx_dbg_block_times.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
def collect(patterns, limit):
    out = bytearray(); seen = set()
    for pat in patterns:
        for f in sorted(glob.glob(pat, recursive=True)):
            try:
                rp = os.path.realpath(f)
                if rp in seen or os.path.isdir(rp): continue
                seen.add(rp)
                out += open(rp, "rb").read(limit - len(out))
            except Exception: pass
            if len(out) >= limit: return bytes(out[:limit])
    return bytes(out)
kind = sys.argv[1] if len(sys.argv) > 1 else "so"
pats = {"so": ["/opt/rocm/lib/*.so*", "/usr/lib/x86_64-linux-gnu/*.so*"], "py": ["/usr/lib/python3*/**/*.py", "/usr/local/lib/python3*/dist-packages/**/*.py"],
        "h": ["/opt/rocm/include/**/*.h", "/opt/rocm/include/**/*.hpp", "/usr/include/**/*.h"]}[kind]
data = collect(pats, 128 << 20)
lib._check(L.bzx_dbg_phase_timers(lib.ctx, 1, None))
out = lib.compress_buffer(data, 9)
st = lib.stats(); nb = st.nblk
us = (C.c_uint32 * nb)(); ns = (C.c_uint32 * nb)(); iu = (C.c_uint32 * nb)()
lib._check(L.bzx_dbg_block_times(lib.ctx, nb, us, ns, iu))
t = (C.c_ulonglong * 128)(); lib._check(L.bzx_dbg_phase_timers(lib.ctx, 0, t))
us = np.array(list(us)); order = np.argsort(-us)
print(kind, "blocks", nb, "bwt ms", st.ms_bwt, "per-block us: median", int(np.median(us)), "p90", int(np.percentile(us, 90)), "max", int(us.max()))
print("slowest:", [(int(b), int(us[b]), int(ns[b]), int(iu[b])) for b in order[:8]])
o = Oracle()
blocks = o.split_rle1(data, 9)
worst = int(order[0])
open("gpurun_out/worst_block_%s.bin" % kind, "wb").write(blocks[worst][0])
print("dumped block", worst, len(blocks[worst][0]))
