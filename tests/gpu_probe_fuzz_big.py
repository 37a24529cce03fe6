"""Ad-hoc GPU probe: differential fuzzing at block-boundary scale (1-4 MB inputs, all levels): long runs across block
boundaries, run-heavy / plain mixtures, inputs that end right at or just past a full block."""
import bz2, random, sys, time
sys.path.insert(0, "tests")
from gpu_probe_fuzz import gen


def big(rnd):
    parts, total = [], rnd.choice([150_000, 1_000_000, 2_700_000, 4_000_000])
    while sum(map(len, parts)) < total:
        k = rnd.randrange(5)
        if k == 0:
            parts.append(gen(rnd))
        elif k == 1:
            parts.append(bytes([rnd.randrange(256)]) * rnd.choice([3, 4, 5, 254, 255, 256, 257, 1000, 70000, 300000]))
        elif k == 2:
            parts.append(rnd.randbytes(rnd.choice([1, 100, 50000, 200000])))
        elif k == 3:
            u = bytes(rnd.choices(b"ab", k=rnd.choice([1, 2, 3, 7])))
            parts.append(u * rnd.choice([10, 1000, 100000]))
        else:
            parts.append(bytes(rnd.choices(b"etaoin shrdlu\n", k=rnd.choice([1000, 400000]))))
    data = b"".join(parts)
    cut = rnd.choice([None, None, 99981, 99982, 199981, 899981, 899982, 899985, 1799962])
    return data[:cut] if cut and cut <= len(data) else data


def main():
    from bzx_ctypes import BzxLib
    lib = BzxLib(max_blocks=64)
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    bad, t0 = 0, time.time()
    for c in range(ncase):
        rnd = random.Random(seed0 * 7919 + c)
        data = big(rnd)
        level = rnd.randrange(1, 10)
        if lib.compress_buffer(data, level) != bz2.compress(data, level):
            bad += 1
            open(f"gpurun_out/fuzzbig_fail_{seed0}_{c}.bin", "wb").write(data)
            print("MISMATCH case", c, len(data), "level", level, flush=True)
        if c % 20 == 19:
            print(f"{c + 1} cases, {bad} mismatches, {time.time() - t0:.0f}s", flush=True)
    print("done", ncase, "cases", bad, "mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
