"""Ad-hoc GPU probe: real-world-ish data from the image (python sources, shared objects): parity vs libbz2 + timing."""
import sys, time, bz2, os, glob
sys.path.insert(0, "tests")
from bzx_ctypes import *
lib = BzxLib(max_blocks=400)
def collect(patterns, limit):
    out = bytearray(); seen = set()
    for pat in patterns:
        for f in sorted(glob.glob(pat, recursive=True)):
            try:
                rp = os.path.realpath(f)
                if rp in seen or os.path.isdir(rp):
                    continue
                seen.add(rp)
                out += open(rp, "rb").read(limit - len(out))
            except Exception:
                pass
            if len(out) >= limit:
                return bytes(out[:limit])
    return bytes(out)
def run(name, data):
    lib.compress_buffer(data[:1 << 20], 9)
    t = time.time(); out = lib.compress_buffer(data, 9); dt = time.time() - t
    st = lib.stats()
    t = time.time(); ref = bz2.compress(data, 9); dr = time.time() - t
    print(f"{name:12s} raw={len(data):10d} out={len(out):10d} ratio={len(out)/max(1,len(data)):.3f} blocks={st.nblk:4d} periodic={st.n_periodic} "
          f"device {len(data)/1e6/(st.ms_total/1e3):8.1f} MB/s (ms split/bwt/mtf/huf/emit={st.ms_split:.1f}/{st.ms_bwt:.1f}/{st.ms_mtf:.1f}/{st.ms_huffman:.1f}/{st.ms_emit:.1f}) "
          f"libbz2 {len(data)/1e6/dr:6.1f} MB/s parity={'OK' if ref == out else 'MISMATCH'}", flush=True)
lim = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 128 << 20
run("py-sources", collect(["/usr/lib/python3*/**/*.py", "/usr/local/lib/python3*/dist-packages/**/*.py"], lim))
run("shared-objs", collect(["/opt/rocm/lib/*.so*", "/usr/lib/x86_64-linux-gnu/*.so*"], lim))
run("headers", collect(["/opt/rocm/include/**/*.h", "/opt/rocm/include/**/*.hpp", "/usr/include/**/*.h"], lim))
