"""Ad-hoc GPU probe: adversarial configs (BASELINE configs[4]) at reduced and full size: parity vs libbz2 + timing."""
import sys, time, bz2, hashlib, ctypes as C
sys.path.insert(0, "tests")
import numpy as np, torch
from bzx_ctypes import *
o = Oracle(); lib = BzxLib(max_blocks=400)
def run(name, data, check=True):
    t = time.time(); out = lib.compress_buffer(data, 9); dt = time.time() - t
    st = lib.stats()
    msg = f"{name:18s} raw={len(data):10d} out={len(out):10d} blocks={st.nblk:4d} periodic={st.n_periodic} wall={dt:7.2f}s ms split/bwt/mtf/huf/emit={st.ms_split:.1f}/{st.ms_bwt:.1f}/{st.ms_mtf:.1f}/{st.ms_huffman:.1f}/{st.ms_emit:.1f}"
    if check:
        t = time.time(); ref = bz2.compress(data, 9); dr = time.time() - t
        msg += f" libbz2={dr:.1f}s parity={'OK' if ref == out else 'MISMATCH'}"
    print(msg, flush=True)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
run("zeros", b"\0" * (mib << 20))
run("random", o.randbytes(mib << 20))
run("text", o.synthtext(mib << 20))
run("ff-runs", (b"\xff" * 1000 + b"abc") * ((mib << 20) // 1003))
