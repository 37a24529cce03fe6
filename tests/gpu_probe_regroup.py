"""Ad-hoc GPU probe: an input with oversized groups (thousands of copies of one string) through the diag build."""
import sys, bz2, random
sys.path.insert(0, "tests")
from bzx_ctypes import *
lib = BzxLib(max_blocks=8)
lib.lib.bzx_dbg_phase_timers.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
rnd = random.Random(5)
ul, pl, nc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
unit = rnd.randbytes(ul)
data = b"".join(unit + rnd.randbytes(pl) for _ in range(nc))
lib._check(lib.lib.bzx_dbg_phase_timers(lib.ctx, 1, None))
out = lib.compress_buffer(data, 9)
t = (C.c_ulonglong * 128)()
lib._check(lib.lib.bzx_dbg_phase_timers(lib.ctx, 0, t))
st = lib.stats()
print(len(data), out == bz2.compress(data, 9), {k: getattr(st, k) for k in ("n_buckets", "n_open_buckets", "n_resume_left", "n_from_scratch")},
      "deep-splits", t[104], "redo-split-nbk", t[106], "redo-split-other", t[107])
