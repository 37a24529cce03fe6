"""Generates tests/golden/streams.json: sha256 + length of libbz2 1.0.8 output (python `bz2`, the C
library the metric names) for seeded synthetic inputs, plus small inline known-answer vectors.
Run in the build container:  python tests/gen_golden.py
The inputs are re-generated from their recipe at test time (oracle/oracle_driver.c generators), so
only hashes are committed."""
import bz2
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

RECIPES = {
    # name: (recipe, level)
    "empty": ({"kind": "bytes", "hex": ""}, 9),
    "silly": ({"kind": "bytes", "hex": b"Making a silly test.".hex()}, 9),
    "config1_text_1MiB_l1": ({"kind": "synthtext", "n": 1 << 20}, 1),
    "config2_text_block_l9": ({"kind": "synthtext", "n": 899981}, 9),
    "text_4MiB_l9": ({"kind": "synthtext", "n": 4 << 20}, 9),
    "text_2MiB_l5": ({"kind": "synthtext", "n": 2 << 20}, 5),
    "random_3MiB_l9": ({"kind": "random", "n": 3 << 20}, 9),
    "zeros_2MiB_l9": ({"kind": "zeros", "n": 2 << 20}, 9),
    "runs_mixed_l9": ({"kind": "runs", "n": 3_000_000, "seed": 17}, 9),
    "allbytes_l1": ({"kind": "cycle256", "n": 300_000}, 1),
    # BASELINE.json configs at full size (sha256 of libbz2's stream; minutes of single-thread libbz2 to regenerate)
    "config3_text_1GiB_l9": ({"kind": "synthtext", "n": 1 << 30}, 9),
    "config5a_zeros_256MiB_l9": ({"kind": "zeros", "n": 256 << 20}, 9),
    "config5b_random_256MiB_l9": ({"kind": "random", "n": 256 << 20}, 9),
}

# Periodic blocks u^k: the 24-bit origPtr is whatever libbz2's sorter leaves among the k identical
# rotations.  (unit recipe, k): unit = bytes.fromhex(hex) or a seeded random unit of `len` bytes over `alpha` symbols.
PERIODIC = [
    ({"zeros": 255}, 2), ({"zeros": 255}, 1999), ({"zeros": 255}, 20000), ({"zeros": 255}, 179996),
    ({"zeros": 1020}, 3), ({"zeros": 259}, 1000),
    ({"hex": "6162"}, 2), ({"hex": "6162"}, 50), ({"hex": "6162"}, 5000), ({"hex": "6162"}, 40000),
    ({"hex": "00000001"}, 3000), ({"hex": "61626361626364"}, 15),
    ({"hex": "61626361626364"}, 1400), ({"hex": "ff"}, 12345), ({"hex": "0001"}, 449990),
    ({"len": 7, "alpha": 3, "seed": 1}, 3), ({"len": 19, "alpha": 2, "seed": 2}, 600), ({"len": 100, "alpha": 4, "seed": 3}, 77),
    ({"len": 1000, "alpha": 200, "seed": 4}, 9), ({"len": 4999, "alpha": 26, "seed": 5}, 2), ({"len": 5000, "alpha": 26, "seed": 6}, 2),
    ({"len": 10007, "alpha": 5, "seed": 7}, 80), ({"len": 280001, "alpha": 26, "seed": 8}, 3), ({"len": 440003, "alpha": 250, "seed": 9}, 2),
    ({"len": 64, "alpha": 2, "seed": 10}, 14000), ({"len": 3, "alpha": 3, "seed": 11}, 299993),
]


def periodic_unit(r):
    if "zeros" in r:
        return b"\0" * r["zeros"]          # RLE1 turns 255 zeros into 00 00 00 00 fb: the image is periodic too
    if "hex" in r:
        return bytes.fromhex(r["hex"])
    import random
    rnd = random.Random(r["seed"])
    syms = rnd.sample(range(256), r["alpha"])
    while True:
        u = bytes(rnd.choice(syms) for _ in range(r["len"]))
        # the unit itself must not be a power of a shorter word, or the recorded k would be wrong
        if not any(r["len"] % p == 0 and u == u[:p] * (r["len"] // p) for p in range(1, r["len"] // 2 + 1) if r["len"] % p == 0):
            return u


def orig_ptr_of_stream(z):
    """origPtr of the FIRST block of a .bz2: 24 bits after 'BZh9' (32) + block magic (48) + CRC (32) + randomised (1)."""
    v = int.from_bytes(z[:20], "big")
    return (v >> (160 - 113 - 24)) & 0xFFFFFF


def make_input(oracle, r):
    k = r["kind"]
    if k == "bytes":
        return bytes.fromhex(r["hex"])
    if k == "synthtext":
        return oracle.synthtext(r["n"])
    if k == "random":
        return oracle.randbytes(r["n"])
    if k == "zeros":
        return b"\0" * r["n"]
    if k == "cycle256":
        return (bytes(range(256)) * (r["n"] // 256 + 1))[:r["n"]]
    if k == "runs":
        # runs of 1..600 of pseudo-random bytes: every RLE1 case incl. 255-splits
        src = oracle.randbytes(r["n"] // 16 + 64, seed=r["seed"])
        out = bytearray()
        i = 0
        while len(out) < r["n"]:
            b, l = src[i], src[i + 1] | ((src[i + 2] & 3) << 8)
            out += bytes([b]) * (1 + l % 600)
            i += 3
        return bytes(out[:r["n"]])
    raise ValueError(k)


def main():
    from bzx_ctypes import Oracle
    o = Oracle()
    # periodic blocks: single-block inputs; the block libbz2 sorts is the RLE1 image of the input (oracle split,
    # itself pinned against libbz2 by the stream tests); origPtr is read from libbz2's stream
    periodic = []
    for unit_r, k in PERIODIC:
        raw = periodic_unit(unit_r) * k
        blocks = o.split_rle1(raw, 9)
        assert len(blocks) == 1, (unit_r, k, len(blocks))
        image = blocks[0][0]
        z = bz2.compress(raw, 9)
        periodic.append({"unit": unit_r, "k": k, "raw_len": len(raw), "n": len(image),
                         "sha256": hashlib.sha256(image).hexdigest(), "orig_ptr": orig_ptr_of_stream(z)})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "periodic.json")
    json.dump({"generator": "tests/gen_golden.py", "libbz2": "1.0.8 (python bz2)", "blocks": periodic},
              open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
    streams = {}
    for name, (recipe, level) in RECIPES.items():
        data = make_input(o, recipe)
        z = bz2.compress(data, level)
        streams[name] = {"input": recipe, "level": level, "raw_len": len(data),
                         "raw_sha256": hashlib.sha256(data).hexdigest(), "bz2_len": len(z),
                         "bz2_sha256": hashlib.sha256(z).hexdigest()}
    kats = {
        # SURVEY.md section 3.4, probed from libbz2 1.0.8
        "empty_l9": {"input_hex": "", "level": 9, "bz2_hex": bz2.compress(b"", 9).hex()},
        "silly_l9": {"input_hex": b"Making a silly test.".hex(), "level": 9,
                     "bz2_hex": bz2.compress(b"Making a silly test.", 9).hex()},
        # reference src/tools/symbol_map.rs:45-59
        "symbol_map_silly": {"input_hex": b"Making a silly test.".hex(), "words": [11008, 32770, 4, 17754, 6208]},
        "symbol_map_full": {"input_hex": bytes(range(256)).hex(), "words": [0xffff] * 17},
        # reference src/bitstream/bitpacker.rs:118-166 (out16/out24/out32 packing)
        "bitpacker": [
            {"name": "out16_test", "ops": [["out16", 0b0010000100100000]], "bytes_hex": b"! ".hex()},
            {"name": "out24_and_loc_test", "ops": [["out24", 0b00001000000000000000000000100001], ["flush", 0],
                                                      ["out24", 0b00011000000000000000000000000011]],
             "bytes_hex": bytes([33, 0, 0, 3]).hex()},
            {"name": "out24_short_test", "ops": [["out24", 0b00000010000000000000000000000011]],
             "bytes_hex": bytes([0b11000000]).hex()},
            {"name": "out32_test", "ops": [["out32", 0b00100001001000000010000100100000]],
             "bytes_hex": bytes([33, 32, 33, 32]).hex()},
        ],
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "streams.json")
    json.dump({"generator": "tests/gen_golden.py", "libbz2": "1.0.8 (python bz2)", "streams": streams, "kats": kats},
              open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
