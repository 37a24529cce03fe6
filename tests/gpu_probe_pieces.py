"""Ad-hoc GPU probe: compress a corpus piece by piece (one bzip2 block each) and save every piece whose stream differs
from libbz2's under gpurun_out/ (to reproduce a mismatch in the CPU emulator).  Usage: gpu_probe_pieces.py hdr|py|so [MiB]"""
import sys, bz2, glob, os
sys.path.insert(0, "tests")
from bzx_ctypes import *
which = sys.argv[1] if len(sys.argv) > 1 else "hdr"
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pats = {"hdr": ["/opt/rocm/include/**/*.h", "/opt/rocm/include/**/*.hpp", "/usr/include/**/*.h"],
        "py": ["/usr/lib/python3*/**/*.py", "/usr/local/lib/python3*/dist-packages/**/*.py"],
        "so": ["/opt/rocm/lib/*.so*", "/usr/lib/x86_64-linux-gnu/*.so*"]}[which]
out = bytearray(); seen = set()
for pat in pats:
    for f in sorted(glob.glob(pat, recursive=True)):
        try:
            rp = os.path.realpath(f)
            if rp in seen or os.path.isdir(rp):
                continue
            seen.add(rp)
            out += open(rp, "rb").read((mib << 20) - len(out))
        except Exception:
            pass
        if len(out) >= mib << 20:
            break
data = bytes(out)
lib = BzxLib(max_blocks=16)
bad = 0
for k, off in enumerate(range(0, len(data), 850000)):
    piece = data[off:off + 850000]
    z = lib.compress_buffer(piece, 9)
    if z != bz2.compress(piece, 9):
        st = lib.stats()
        bad += 1
        print("MISMATCH piece", k, "offset", off, len(piece), "gave-up", st.n_open_buckets, "left", st.n_open_left,
              "blocks-left", st.n_resume_left, flush=True)
        if bad <= 3:
            open(f"gpurun_out/bad_piece_{which}_{k}.bin", "wb").write(piece)
print("pieces", k + 1, "bad", bad)
