"""Ad-hoc GPU probe of the bucket sorter: parity vs libbz2 / the oracle, stage times, buckets, blocks handed over."""
import sys, time, bz2, os, glob
sys.path.insert(0, "tests")
from bzx_ctypes import *
o = Oracle()
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
DIAG = "diag" in os.environ.get("BZX_LIB", "")
SORT_SLOTS = {64: "fetch", 65: "load", 66: "sort32", 67: "sort32+flags", 73: "list", 68: "w-build+list", 69: "big-groups", 70: "tiny-rank",
              71: "tiny-write", 72: "output"}
SPLIT_SLOTS = {96: "fetch", 97: "in-use", 98: "pack", 99: "hist", 100: "form", 101: "partition", 102: "emit", 103: "deeper"}
COUNTS = {80: "rounds", 88: "tied-entries", 83: "buckets-done", 84: "fail-rounds", 85: "fail-depth", 86: "medium-groups", 87: "large-groups",
          104: "deep-splits", 105: "deep-split-records", 106: "redo-split-nbk", 107: "redo-split-other"}
if DIAG:
    lib.lib.bzx_dbg_phase_timers.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
def run(name, data, ref="oracle"):
    lib.compress_buffer(data[:1 << 20], 9)
    if DIAG:
        lib._check(lib.lib.bzx_dbg_phase_timers(lib.ctx, 1, None))
    out = lib.compress_buffer(data, 9)
    st = lib.stats()
    if DIAG:
        t = (C.c_ulonglong * 128)()
        lib._check(lib.lib.bzx_dbg_phase_timers(lib.ctx, 0, t))
        tot = sum(t[i] for i in SORT_SLOTS) or 1
        print("   sort kernel (share of workgroup time): " + "  ".join(f"{SORT_SLOTS[i]} {100.0*t[i]/tot:.1f}%" for i in SORT_SLOTS),
              f" [sum {tot/100.0/512:.0f} us per workgroup slot]")
        tot = sum(t[i] for i in SPLIT_SLOTS) or 1
        print("   split kernel: " + "  ".join(f"{SPLIT_SLOTS[i]} {100.0*t[i]/tot:.1f}%" for i in SPLIT_SLOTS),
              f" [sum {tot/100.0/max(1,st.nblk):.0f} us per block]")
        print("   counts: " + "  ".join(f"{v}={t[k]}" for k, v in COUNTS.items()))
        print(f"   rank rounds: bucket-rounds by workgroups {t[110]} (entries {t[112]}), by single waves {t[111]}; open buckets per round:", [t[113 + r] for r in range(15)])
        nb = max(1, st.nblk)
        print("   huffman (us per block): " + "  ".join(f"{v} {t[k]/100.0/nb:.0f}" for k, v in
              {40: "setup", 41: "costs+freq x4", 42: "code-lengths x4", 43: "codes", 44: "payload-bits", 45: "selector-mtf"}.items()))
        print("   mtf (us per block): " + "  ".join(f"s{k} {t[k]/100.0/nb:.0f}" for k in range(32, 40)))
        print("   emit (us per block): " + "  ".join(f"s{k} {t[k]/100.0/nb:.0f}" for k in range(48, 52)))
        gen = {0: "A+hist", 1: "lsd4", 2: "rerank0", 3: "tail", 4: "rank-rerank", 5: "isa-build", 6: "rank-segsort", 7: "rank-bigsplit"}
        tg = sum(t[i] for i in range(32)) or 1
        print("   general sorter: " + "  ".join(f"{v} {t[k]/100.0/max(1,st.n_redo):.0f}us" for k, v in gen.items()),
              f" text-rounds {sum(t[i] for i in range(8, 32))/100.0/max(1,st.n_redo):.0f}us  [per block handed over]",
              " rank rounds m/maxgrp:", [(t[52 + 2 * r], t[53 + 2 * r]) for r in range(6)])
    t = time.time()
    want = o.compress_mt(data, 9) if ref == "oracle" else bz2.compress(data, 9)
    dr = time.time() - t
    print(f"{name:12s} raw={len(data):10d} ratio={len(out)/max(1,len(data)):.3f} blocks={st.nblk:4d} buckets={st.n_buckets:6d} "
          f"redo={st.n_redo:3d} periodic={st.n_periodic} host-to-host {len(data)/1e6/(st.ms_total/1e3):7.1f} MB/s, stage kernels "
          f"{len(data)/1e6/((st.ms_split+st.ms_bwt+st.ms_mtf+st.ms_huffman+st.ms_emit)/1e3):7.1f} MB/s "
          f"ms split/bwt[split,sort,general]/mtf/huf/emit={st.ms_split:.2f}/{st.ms_bwt:.2f}[{st.ms_bwt_split:.2f},{st.ms_bwt_sort:.2f},"
          f"{st.ms_bwt_general:.2f} (rank {st.ms_bwt_rank:.2f}, gave-up {st.n_open_buckets} left {st.n_open_left} blocks-left {st.n_resume_left} from-scratch {st.n_from_scratch})]/{st.ms_mtf:.2f}/{st.ms_huffman:.2f}/{st.ms_emit:.2f} ref {dr:.1f}s parity={'OK' if want == out else 'MISMATCH'}",
          flush=True)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["text", "one", "random", "py", "so", "hdr", "zeros"]
if "one" in which: run("one-block", o.synthtext(899981))
if "text" in which: run("text", o.synthtext(mib << 20))
if "random" in which: run("random", o.randbytes(mib << 20))
if "py" in which: run("py-sources", collect(["/usr/lib/python3*/**/*.py", "/usr/local/lib/python3*/dist-packages/**/*.py"], mib << 20))
if "so" in which: run("shared-objs", collect(["/opt/rocm/lib/*.so*", "/usr/lib/x86_64-linux-gnu/*.so*"], mib << 20))
if "hdr" in which: run("headers", collect(["/opt/rocm/include/**/*.h", "/opt/rocm/include/**/*.hpp", "/usr/include/**/*.h"], mib << 20))
if "zeros" in which: run("zeros", b"\0" * (mib << 20), ref="bz2")
