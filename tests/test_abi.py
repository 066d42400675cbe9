"""The C ABI: the shared library loads, exports every function include/vdl2hip.h declares, and
fails loudly (no silent CPU path) when there is no GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from dumpvdl2_amd import build, vdl2hip
    build.build()
    return vdl2hip.load_library()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "vdl2hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vdl2hip_[a-z_0-9]+)\s*\(", src)) - {"vdl2hip_frame_cb"})


def test_every_declared_symbol_is_exported(lib):
    from dumpvdl2_amd import vdl2hip
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vdl2hip.h but not exported"
    assert sorted(vdl2hip.EXPORTS) == names
    assert lib.vdl2hip_abi_version() == 6


def test_struct_layouts_match_header(lib):
    from dumpvdl2_amd import vdl2hip
    # sizes the C compiler gives the structs of the header (checked with a tiny C program)
    import subprocess, tempfile
    prog = '#include <stdio.h>\n#include "vdl2hip.h"\nint main(){printf("%zu %zu %zu %d\\n",sizeof(vdl2hip_cfg),sizeof(vdl2hip_frame),sizeof(vdl2hip_stats),VDL2HIP_NUM_COUNTERS);return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        a, b, c, n = map(int, subprocess.check_output([os.path.join(d, "t")]).split())
    assert (a, b, c, n) == (C.sizeof(vdl2hip.Cfg), C.sizeof(vdl2hip.CFrame), C.sizeof(vdl2hip.Stats), vdl2hip.NUM_COUNTERS)


def test_bad_arguments_are_rejected(lib):
    from dumpvdl2_amd import vdl2hip
    h = C.c_void_p()
    assert lib.vdl2hip_create(None, C.byref(h)) == -1
    cfg = vdl2hip.Cfg(C.sizeof(vdl2hip.Cfg), 136975000, 0, 1, 1, (C.c_uint32 * 1)(136975000), 0.0, 0, 0, 0, 0)
    assert lib.vdl2hip_create(C.byref(cfg), C.byref(h)) == -1          # oversample 0
    assert lib.vdl2hip_strerror(-3).decode() == "HIP device error"
    assert lib.vdl2hip_feed(None, None, 0) == -1


def test_no_gpu_means_error_not_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from dumpvdl2_amd import vdl2hip
    with pytest.raises(vdl2hip.Vdl2HipError, match="HIP device error"):
        vdl2hip.Receiver(136975000, [136975000], 10)


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing that ships (package, C ABI sources, headers, tools) may import, include, link or
    execute it - only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline / parity gate do."""
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for sub in ("dumpvdl2_amd", "include", "tools"):
        for dp, _, files in os.walk(os.path.join(root, sub)):
            for f in files:
                if not f.endswith((".py", ".h", ".hip", ".c", ".cpp")):
                    continue
                txt = open(os.path.join(dp, f), errors="replace").read()
                for m in re.finditer(r"^.*(import\s+oracle|from\s+oracle|oracle/|libvdl2oracle|vdl2o_|pyoracle).*$", txt, re.M):
                    if "test-only" in m.group(0) or "tests" in m.group(0):
                        continue
                    bad.append((os.path.relpath(os.path.join(dp, f), root), m.group(0).strip()[:100]))
    assert not bad, bad


def test_library_source_builds_without_rccl_headers():
    """RCCL is optional at build time as well as at run time: the host side of vdl2hip.hip must compile with the fallback
    declarations of group.inc (an installation without rccl/rccl.h still gets the single-GPU library and the peer-copy group)."""
    import shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "--cuda-host-only", "-DVDL2HIP_NO_RCCL_HEADER",
                        os.path.join(ROOT, "dumpvdl2_amd", "csrc", "vdl2hip.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
