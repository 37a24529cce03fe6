"""CPU test (-m "not gpu") of the N>1 path: world sizes 2 and 4 over gloo on 127.0.0.1.
The processes run bzx_shard_prepare -> all_reduce(block sizes) -> bzx_shard_emit_packed -> gather ->
bzx_shard_assemble_begin/_rank on rank 0 exactly as bench.py does with RCCL, but through the kernel emulator
(tests/emu) with CPU tensors; rank 0's result must be byte-identical to libbz2's stream of the whole input.
Inputs: run-free text (zero-copy blocks) and a run-heavy mix (zeros + runs of 1..600: RLE1 changes every block and
runs cross block borders -- SURVEY.md 8f N3: the split of general input before the blocks are dealt to the ranks)."""
import os
import socket
import subprocess
import sys

import pytest

from bzx_ctypes import EMU_PATH, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,kind,nbytes,mode", [(2, "text", 230000, "whole"), (4, "runs", 900000, "whole"),
                                                    (4, "runs", 900000, "scan"), (2, "dups", 420000, "whole"),
                                                    (2, "border", 450000, "scan"), (4, "border", 600000, "scan")])
def test_round_robin_sharding(oracle, world, kind, nbytes, mode):
    # mode "scan": the sharded split analysis (bzx_shard_scan_* + all-gathers of 24 B per tile), with runs of more than
    # 255 bytes straddling the borders between the ranks' tile shares ("border")
    srcs = [os.path.join(ROOT, "bzip2-rust_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "bzip2-rust_amd", "csrc"))]
    if not os.path.exists(EMU_PATH) or any(os.path.getmtime(s) > os.path.getmtime(EMU_PATH) for s in srcs):
        subprocess.check_call(["bash", os.path.join(ROOT, "tests", "emu", "build_emu.sh")])
    port = str(_free_port())
    worker = os.path.join(ROOT, "tests", "shard_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), port, str(nbytes), kind, mode],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=900)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARD_OK" in outs[0], outs[0]
    nblk = int(outs[0].split("SHARD_OK")[1].split()[0])
    assert nblk >= world, outs[0]          # every rank owns at least one block
