"""Ad-hoc GPU probe: run the BWT kernel truncated after phase k (k = 1..6, then complete) so that per-dispatch
PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE) give the HBM traffic of each phase by difference."""
import sys, ctypes as C
sys.path.insert(0, "tests")
from bzx_ctypes import *
o = Oracle(); lib = BzxLib(sys.argv[2]) if len(sys.argv) > 2 else BzxLib()
L = lib.lib
L.bzx_dbg_time_stages.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_int, C.POINTER(C.c_float)]
L.bzx_dbg_set_stop.argtypes = [C.c_void_p, C.c_uint32]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1194
blk = o.synthtext(899981)
ms = (C.c_float * 4)()
stops = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1, 2, 3, 4, 5, 6, 0]
lib._check(L.bzx_dbg_set_stop(lib.ctx, stops[0]))
lib._check(L.bzx_dbg_time_stages(lib.ctx, blk, len(blk), reps, 1, ms))      # warm-up
names = ["complete", "I1 hist", "+I2 radix", "+R0 rerank", "+big split r0", "+seg sort r0", "+rerank r0"]
for k in stops:
    lib._check(L.bzx_dbg_set_stop(lib.ctx, k))
    lib._check(L.bzx_dbg_time_stages(lib.ctx, blk, len(blk), reps, 1, ms))
    print(f"stop={k} {names[k]:16s} bwt {ms[0]:8.2f} ms", flush=True)
