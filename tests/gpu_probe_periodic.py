"""Ad-hoc GPU probe: periodic blocks one at a time, with timing; run it under `timeout`."""
import sys, time, json
sys.path.insert(0, "tests")
from bzx_ctypes import *
from gen_golden import periodic_unit
o = Oracle(); lib = BzxLib(max_blocks=16)
fx = json.load(open("tests/golden/periodic.json"))
lim = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
for g in sorted(fx["blocks"], key=lambda g: g["n"]):
    if g["n"] > lim: continue
    raw = periodic_unit(g["unit"]) * g["k"]
    image = o.split_rle1(raw, 9)[0][0]
    print("start", g["unit"], g["k"], g["n"], flush=True)
    t = time.time(); L, orig, status = lib.stage_bwt(image); dt = time.time() - t
    print(f"   {dt*1e3:9.1f} ms  orig {orig} want {g['orig_ptr']} {'OK' if orig == g['orig_ptr'] else 'MISMATCH'} status {status}", flush=True)
