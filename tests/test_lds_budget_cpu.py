"""No kernel of the product library may hold more than 64 KB of STATIC LDS. Round 5 found why the hard way: stream_attn_kernel (78 KB of static LDS, small enough to share a CU)
read wrong data beside another session's kernels, silently (profiles/r05_stream_determinism.txt). Kernels that need more declare DYNAMIC LDS through hipFuncSetAttribute and either
own their CU (block / cluster kernels) or were checked for determinism beside a co-tenant. The check reads the gfx950 code object out of the built library's offload bundle."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "automatic-speech-recognition-asr-onnx_amd", "libasr_mi355x.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob):
    """(triple, bytes) of every entry of every clang offload bundle inside `blob`"""
    out, at = [], blob.find(MAGIC)
    while at >= 0:
        n, = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            out.append((triple, blob[at + off:at + off + size]))
            p += 24 + tl
        at = blob.find(MAGIC, at + len(MAGIC))
    return out


@pytest.mark.skipif(not (os.path.isfile(LIB) and os.path.isfile(READELF)), reason="needs the built library and llvm-readelf")
def test_no_kernel_holds_more_than_64_kb_of_static_lds():
    blob = open(LIB, "rb").read()
    objs = [(t, b) for t, b in _code_objects(blob) if "gfx950" in t and b[:4] == b"\x7fELF"]
    assert objs, "no gfx950 code object in the library"
    kernels = {}
    for _, elf in objs:
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        pending = None                        # (a kernel's keys come sorted: .args (with their own .name lines), .group_segment_fixed_size, ..., .name, ...)
        for line in notes.splitlines():
            m = re.search(r"\.group_segment_fixed_size:\s+(\d+)", line)
            if m:
                pending = int(m.group(1))
                continue
            m = re.search(r"\.name:\s+(\S+)", line)
            if m and pending is not None:
                kernels[m.group(1)] = pending
                pending = None
    assert len(kernels) > 100, len(kernels)                      # (the library has a few hundred kernel instances)
    # no exemption (round 6): the f32 instance of stream_attn_kernel keeps its K / V images in DYNAMIC LDS and is launched with the rest of the CU on top
    over = {k: v for k, v in kernels.items() if v > 64 * 1024}
    assert not over, over
    assert any("stream_attn_kernelIfE" in k and v <= 16 * 1024 for k, v in kernels.items())
    assert any("stream_attn_kernelItE" in k and v <= 48 * 1024 for k, v in kernels.items())
    assert max(kernels.values()) > 40 * 1024                      # (sanity: the parser really reads sizes -- stream_attn_kernel<bf16> holds 45 696 B)
