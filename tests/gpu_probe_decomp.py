"""Ad-hoc GPU probe: device decompression throughput (device-resident in and out) for text, random and zeros."""
import sys, time, ctypes as C
sys.path.insert(0, "tests")
import torch
from bzx_ctypes import *
o = Oracle(); lib = BzxLib(max_blocks=400); L = lib.lib
L.bzx_decompress_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for name, data in (("text", o.synthtext(mib << 20)), ("random", o.randbytes(mib << 20)), ("zeros", b"\0" * (mib << 20))):
    z = lib.compress_buffer(data, 9)
    d_z = torch.frombuffer(bytearray(z), dtype=torch.uint8).cuda()
    d_o = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda")
    ol = C.c_size_t()
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        lib._check(L.bzx_decompress_device(lib.ctx, d_z.data_ptr(), len(z), d_o.data_ptr(), len(data) + 64, C.byref(ol)))
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    ok = bytes(d_o[:ol.value].cpu().numpy()) == data
    print(f"{name:8s} {len(z):11d} -> {ol.value:11d} bytes in {dt*1e3:8.1f} ms = {ol.value/1e6/dt:8.1f} MB/s  {'OK' if ok else 'MISMATCH'}", flush=True)
