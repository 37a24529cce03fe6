"""Ad-hoc GPU probe: differential fuzzing of the whole device path against libbz2 on structured random inputs
(alphabet sizes, repeats at several scales, record-like data, runs, near-periodic data, mixtures)."""
import bz2, random, sys, time
sys.path.insert(0, "tests")


def gen(rnd):
    k = rnd.choice([1, 2, 3, 4, 7, 16, 27, 64, 100, 200, 256])
    syms = rnd.sample(range(256), k)
    w = [1.0 / (i + 1) ** rnd.choice([0, 0.5, 1, 2]) for i in range(k)]
    def noise(n):
        return bytes(rnd.choices(syms, w, k=n))
    kind = rnd.randrange(9)
    n = rnd.choice([3000, 40000, 150000, 420000])
    if kind == 0:
        return noise(n)
    if kind == 1:                                   # copies with mutations
        base = bytearray(noise(rnd.choice([500, 5000, 60000])))
        out = bytearray()
        while len(out) < n:
            c = bytearray(base)
            for _ in range(rnd.randrange(0, 20)):
                c[rnd.randrange(len(c))] = rnd.choice(syms)
            out += c
        return bytes(out)
    if kind == 2:                                   # fixed-size records with a varying field
        rec = noise(rnd.choice([8, 40, 190]))
        return b"".join(rec + bytes([syms[i % k], syms[(i * 7) % k]]) for i in range(n // (len(rec) + 2)))
    if kind == 3:                                   # many occurrences of a few prefixes, distinct continuations
        pres = [noise(rnd.choice([3, 4, 5, 8])) for _ in range(rnd.choice([1, 5, 40]))]
        parts = [rnd.choice(pres) + noise(rnd.choice([1, 2, 6])) for _ in range(n // 8)]
        return b"".join(parts)
    if kind == 4:                                   # runs
        out = bytearray()
        while len(out) < n:
            out += bytes([rnd.choice(syms)]) * rnd.choice([1, 2, 3, 4, 5, 30, 300, 5000])
        return bytes(out)
    if kind == 5:                                   # periodic / nearly periodic
        u = noise(rnd.choice([1, 2, 5, 17, 1000]))
        d = bytearray(u * (n // len(u) + 1))[:n]
        for _ in range(rnd.choice([0, 0, 1, 5])):
            d[rnd.randrange(len(d))] = rnd.choice(syms)
        return bytes(d)
    if kind == 6:                                   # x + x (every rotation has a twin)
        x = noise(n // 2)
        return x + x + noise(rnd.randrange(0, 4))
    if kind == 7:                                   # mixture
        return gen(rnd)[: n // 2] + gen(rnd)[: n // 2]
    return noise(rnd.randrange(1, 20))              # tiny


def main():
    from bzx_ctypes import BzxLib
    lib = BzxLib(max_blocks=64)
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    bad = 0
    t0 = time.time()
    for c in range(ncase):
        rnd = random.Random(seed0 * 100003 + c)
        data = gen(rnd)
        level = rnd.choice([1, 1, 2, 9])
        got = lib.compress_buffer(data, level)
        if got != bz2.compress(data, level):
            bad += 1
            open(f"gpurun_out/fuzz_fail_{seed0}_{c}.bin", "wb").write(data)
            print("MISMATCH case", c, len(data), "level", level, flush=True)
        if c % 25 == 24:
            print(f"{c + 1} cases, {bad} mismatches, {time.time() - t0:.0f}s", flush=True)
    print("done", ncase, "cases", bad, "mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
