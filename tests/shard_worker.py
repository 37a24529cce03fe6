"""Worker of tests/test_shard_gloo.py: one rank of the round-robin block sharding over gloo.
Runs the REAL C ABI (bzx_shard_prepare / bzx_shard_emit_packed / bzx_shard_assemble_*) through the CPU kernel
emulator (tests/emu), with CPU tensors standing in for HBM, so the N>1 control flow is exercised without GPUs."""
import bz2
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bzx_ctypes import EMU_PATH, BzxLib, Oracle  # noqa: E402


def main():
    rank, world, port, nbytes = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    kind = sys.argv[5] if len(sys.argv) > 5 else "text"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    level = 1
    o = Oracle()
    if kind == "runs":
        from gen_golden import make_input
        # zeros (one long run across several blocks' worth of input), then runs of 1..600, then text
        mix = b"\0" * (nbytes // 3) + make_input(o, {"kind": "runs", "n": nbytes // 3, "seed": 5}) + o.synthtext(nbytes // 3)
        data = np.frombuffer(mix, dtype=np.uint8).copy()
    elif kind == "dups":
        # duplicated stretches inside and across blocks: buckets give up, the fill pass and the rank rounds run on
        # descriptors indexed by the global block number and slabs indexed by the local one
        t = o.synthtext(nbytes // 2)
        mix = t[: nbytes // 4] + t[1000:40000] + t[nbytes // 4:] + t[2000:30000] + t[: nbytes // 3]
        data = np.frombuffer(mix[:nbytes], dtype=np.uint8).copy()
    elif kind == "border":
        # runs longer than 255 that straddle the borders between the ranks' tile shares of the sharded analysis, a run
        # that ends exactly at a border and one that starts there (tiles are 8 KiB)
        data = np.frombuffer(o.synthtext(nbytes), dtype=np.uint8).copy()
        per = ((nbytes + 8191) // 8192 + 2 + world - 1) // world
        for r in range(1, world):
            b = r * per * 8192
            if b + 2000 < nbytes:
                data[b - 300:b + 400] = 0xAA
                if r % 2 == 0:
                    data[b - 300:b] = 0x55
                    data[b:b + 260] = 0x56
    else:
        data = np.frombuffer(o.synthtext(nbytes), dtype=np.uint8).copy()
    raw = np.zeros(data.nbytes + 64, dtype=np.uint8)
    off = (-raw.ctypes.data) % 16
    raw[off:off + data.nbytes] = data
    lib = BzxLib(EMU_PATH)
    L = lib.lib
    L.bzx_shard_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_uint32,
                                    C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t]
    L.bzx_shard_emit_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_size_t)]
    L.bzx_shard_assemble_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.bzx_shard_assemble_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    bits = torch.zeros(64, dtype=torch.int64)
    nblk = C.c_uint32()
    if len(sys.argv) > 6 and sys.argv[6] == "scan":
        # sharded split analysis: the per-byte passes on this rank's share of the tiles, 24 bytes per
        # tile all-gathered, then the chain of boundaries on every rank
        L.bzx_shard_scan_entries.restype = C.c_size_t
        L.bzx_shard_scan_entries.argtypes = [C.c_size_t, C.c_uint32]
        for f in (L.bzx_shard_scan_runs, L.bzx_shard_scan_counts):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p]
        L.bzx_shard_prepare_scanned.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p,
                                                C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t]
        P = L.bzx_shard_scan_entries(data.nbytes, world)
        tiles = torch.full((3, world, P), -7, dtype=torch.int64)          # (garbage where nobody writes)
        lib._check(L.bzx_shard_scan_runs(lib.ctx, raw.ctypes.data + off, data.nbytes, rank, world, tiles.data_ptr()))
        lib._check(L.bzx_ctx_sync(lib.ctx))
        dist.all_gather_into_tensor(tiles[0].view(-1), tiles[0, rank].clone())
        lib._check(L.bzx_shard_scan_counts(lib.ctx, raw.ctypes.data + off, data.nbytes, rank, world, tiles.data_ptr()))
        lib._check(L.bzx_ctx_sync(lib.ctx))
        for a in (1, 2):
            dist.all_gather_into_tensor(tiles[a].view(-1), tiles[a, rank].clone())
        lib._check(L.bzx_shard_prepare_scanned(lib.ctx, raw.ctypes.data + off, data.nbytes, level, rank, world,
                                               tiles.data_ptr(), C.byref(nblk), bits.data_ptr(), bits.numel()))
    else:
        lib._check(L.bzx_shard_prepare(lib.ctx, raw.ctypes.data + off, data.nbytes, level, rank, world, C.byref(nblk),
                                       bits.data_ptr(), bits.numel()))
    mine = bits.clone()
    dist.all_reduce(bits)
    # every block size is reported by exactly one rank
    owners = torch.zeros(64, dtype=torch.int64)
    owners[:nblk.value] = (mine[:nblk.value] != 0).to(torch.int64)
    dist.all_reduce(owners)
    assert bool((owners[:nblk.value] == 1).all()), owners
    assert all(int(mine[b]) != 0 for b in range(rank, nblk.value, world))
    cap = data.nbytes + 65536
    packed = torch.zeros(cap // 4, dtype=torch.int32)
    pl, sl = C.c_size_t(), C.c_size_t()
    lib._check(L.bzx_shard_emit_packed(lib.ctx, bits.data_ptr(), packed.data_ptr(), cap, C.byref(pl), C.byref(sl)))
    assert pl.value % 4 == 0
    # the common length of the gather: the library knows the longest packed buffer of any rank from the sizes;
    # checked here against the collective it replaces
    L.bzx_shard_packed_max.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    pm = C.c_size_t()
    lib._check(L.bzx_shard_packed_max(lib.ctx, C.byref(pm)))
    pmax = torch.tensor([pl.value], dtype=torch.int64)
    dist.all_reduce(pmax, op=dist.ReduceOp.MAX)
    assert pm.value == int(pmax.item()), (pm.value, int(pmax.item()))
    n4 = pm.value // 4
    parts = [torch.zeros(n4, dtype=torch.int32) for _ in range(world)] if rank == 0 else None
    dist.gather(packed[:n4].contiguous(), parts, dst=0)      # every compressed byte travels once
    if rank == 0:
        out = torch.full((cap // 4,), -1, dtype=torch.int32)   # assemble_begin must clear what it uses
        ol = C.c_size_t()
        lib._check(L.bzx_shard_assemble_begin(lib.ctx, out.data_ptr(), cap, C.byref(ol)))
        assert ol.value == sl.value
        for r in range(world):
            lib._check(L.bzx_shard_assemble_rank(lib.ctx, parts[r].data_ptr(), r, out.data_ptr()))
        lib._check(L.bzx_ctx_sync(lib.ctx))
        z = out.numpy().tobytes()[:ol.value]
        want = bz2.compress(data.tobytes(), level)
        assert z == want, (len(z), len(want))
        assert nblk.value == len(o.split_rle1(data.tobytes(), level))
        print("SHARD_OK", nblk.value, len(z))
    dist.barrier()
    dist.destroy_process_group()
    lib.close()


if __name__ == "__main__":
    main()
