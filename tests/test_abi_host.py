"""CPU tests (-m "not gpu"): the C-ABI library loads, exports every symbol include/bzx.h declares,
its host-only pieces (stream assembler, error paths) behave, and it has no CPU compute path."""
import bz2
import ctypes as C
import os
import re

import pytest

from bzx_ctypes import LIB_PATH, ROOT


def _lib():
    if not os.path.exists(LIB_PATH):
        import sys
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return C.CDLL(LIB_PATH)


def test_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "bzx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(bzx_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 18
    lib = _lib()
    for n in sorted(names):
        assert hasattr(lib, n), n


def test_version_and_strerror():
    lib = _lib()
    lib.bzx_version.restype = C.c_char_p
    lib.bzx_strerror.restype = C.c_char_p
    assert b"gfx950" in lib.bzx_version()
    assert lib.bzx_strerror(0) == b"ok"
    assert b"device" in lib.bzx_strerror(-1)


def test_no_cpu_fallback_without_device():
    """There is no GPU in the build container: context creation must fail loudly, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _lib()
    ctx = C.c_void_p()
    rc = lib.bzx_ctx_create(0, 16, C.byref(ctx))
    assert rc == -1 and not ctx.value      # BZX_E_NODEVICE
    from bzx_ctypes import BzxLib, BzxError
    with pytest.raises(BzxError):
        BzxLib()


def test_stream_assembler_matches_libbz2(oracle):
    """bzx_stream_* (BitWriter, bitwriter.rs:42-172) fed with oracle block images == libbz2 stream."""
    lib = _lib()
    lib.bzx_stream_begin.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.bzx_stream_append_block.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint8]
    lib.bzx_stream_finish.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    lib.bzx_stream_free.argtypes = [C.c_void_p]
    for data, level in ((oracle.synthtext(350000), 1), (b"", 9), (b"Making a silly test.", 9),
                        (b"\0" * 1000 + oracle.randbytes(150000), 1)):
        s = C.c_void_p()
        assert lib.bzx_stream_begin(level, C.byref(s)) == 0
        for blk, crc in oracle.split_rle1(data, level):
            img, pad = oracle.compress_block(blk, crc)
            assert lib.bzx_stream_append_block(s, img, len(img), pad) == 0
        p = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        assert lib.bzx_stream_finish(s, C.byref(p), C.byref(n)) == 0
        out = bytes(p[:n.value])
        lib.bzx_stream_free(s)
        assert out == bz2.compress(data, level)
    s = C.c_void_p()
    assert lib.bzx_stream_begin(0, C.byref(s)) == -2    # BZX_E_PARAM


def test_command_line_tool_builds_and_refuses_to_run_without_a_device():
    """tools/bzx.cpp is host glue over the C ABI: it must exist after build() and, like the library,
    it has no CPU path."""
    import subprocess
    exe = os.path.join(ROOT, "bzip2-rust_amd", "bzx")
    if not os.path.exists(exe):
        subprocess.check_call(["bash", os.path.join(ROOT, "bzip2-rust_amd", "build.sh")])
    assert b"bzx" in subprocess.run([exe, "--version"], stdout=subprocess.PIPE, check=True).stdout
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        r = subprocess.run([exe, "-c"], input=b"abc", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 2 and b"no CPU path" in r.stderr
