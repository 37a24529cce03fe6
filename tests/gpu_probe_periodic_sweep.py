"""Ad-hoc GPU probe: exactly periodic inputs u^k of many shapes (alphabet, unit length, copies, size) against live libbz2
(tie order among identical rotations); run it under `timeout`.  Usage: gpu_probe_periodic_sweep.py [seed] [cases]"""
import sys, bz2, random, time
sys.path.insert(0, "tests")
from bzx_ctypes import *
lib = BzxLib(max_blocks=16)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rnd = random.Random(seed)
bad = 0; t0 = time.time(); nper = 0
for c in range(ncase):
    a = rnd.choice([1, 2, 2, 3, 5, 16])
    syms = rnd.sample(range(256), a)
    ell = rnd.choice([1, 2, 3, 4, 5, 7, 12, 31, 64, 100, 257, 1000, 4099, 30011])
    unit = bytes(rnd.choice(syms) for _ in range(ell))
    n = rnd.choice([9000, 20000, 120000, 450000, 899000])
    k = max(2, n // ell)
    data = unit * k
    if rnd.random() < 0.3:
        data = data + unit[: rnd.randrange(0, ell + 1)]         # not a whole number of copies: near-periodic
    lvl = rnd.choice([1, 5, 9])
    t = time.time()
    z = lib.compress_buffer(data, lvl)
    nper += lib.stats().n_periodic
    ok = z == bz2.compress(data, lvl)
    if not ok:
        bad += 1
        print("MISMATCH", c, "alphabet", a, "unit", ell, "copies", k, "len", len(data), "level", lvl, flush=True)
    if time.time() - t > 3:
        print(f"  slow: case {c} unit {ell} x {k} len {len(data)} level {lvl}: {time.time()-t:.1f}s", flush=True)
print(f"done {ncase} cases, {bad} mismatches, {nper} periodic blocks (last calls), {time.time()-t0:.0f}s")
sys.exit(1 if bad else 0)
