"""Build libvdl2hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950."""
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvdl2hip.so")
SOURCES = ["vdl2hip.hip", "group.inc", "ubench.inc", "kernels.h", "vdl2_core.h", "design.h", "tables.h"]
# -ffp-contract=off: the walker/burst code must keep the reference's mul/add sequence;
# the channeliser asks for FMAs explicitly where it wants them.
# -fno-slp-vectorize: packed FP32 issues at half rate on CDNA4, so a v_pk_add the SLP vectoriser glues together from two scalar
# adds gains nothing and costs the moves that line its operands up (sync screening kernel 2.05 -> 1.71 ms at 256 channels, the
# channeliser 1 % faster; gpurun_out r02q).  Where the channeliser wants packed operations it writes float2 arithmetic itself.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-Wall"]


def hipcc_path():
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: libvdl2hip.so cannot be built (there is no CPU fallback)")
    return p


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(os.path.dirname(HERE), "include", "vdl2hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    # (-save-temps: the device assembly comes out as a by-product, for the check below)
    tmp = tempfile.mkdtemp(prefix="vdl2hip_build_")
    try:
        cmd = [hipcc_path()] + FLAGS + ["-save-temps=obj", "-o", os.path.join(tmp, "libvdl2hip.so"), os.path.join(CSRC, "vdl2hip.hip")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=tmp)
        # No out-of-line device function calls: every kernel of this library is one straight piece of code.  When the walker's call
        # tree (walk_run & co.) was left to the inliner and grew past its patience, the kernels that called it faulted on the device
        # (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION) - with nothing to see on the CPU build.  The functions concerned are
        # always_inline; this is the guard for the next one.
        calls = []
        for f in os.listdir(tmp):
            if f.endswith(".s") and "amdgcn" in f:
                with open(os.path.join(tmp, f), errors="replace") as fh:
                    calls += [ln.strip() for ln in fh if "s_swappc_b64" in ln]
        if calls:
            raise RuntimeError(f"the device code calls {len(calls)} function(s) out of line (mark them always_inline): {calls[:3]}")
        shutil.move(os.path.join(tmp, "libvdl2hip.so"), LIB)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return LIB


def build_dropin(force=False):
    """Compile the reference-named C adapter (csrc/dropin.c) to an object a dumpvdl2 tree (or the test harness) links."""
    src = os.path.join(CSRC, "dropin.c")
    obj = os.path.join(HERE, "vdl2hip_dropin.o")
    inc = os.path.join(os.path.dirname(HERE), "include")
    if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(inc, "vdl2hip_dropin.h"))):
        subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-Wall", "-Wextra", "-fPIC", "-I", inc, "-c", src, "-o", obj])
    return obj


def build_harness(out_path):
    """Link tests/dropin_harness.c (stand-in for the unmodified dumpvdl2 main) against the adapter and libvdl2hip.so."""
    root = os.path.dirname(HERE)
    subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-Wall", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "dropin_harness.c"), build_dropin(), "-L", HERE, "-lvdl2hip",
                           "-Wl,-rpath," + HERE, "-lpthread", "-o", out_path])
    return out_path


def build_cli(out_path=None):
    """tools/vdl2hip_iqfile: the `dumpvdl2 --iq-file` work-alike (C) on top of libvdl2hip.so."""
    root = os.path.dirname(HERE)
    out_path = out_path or os.path.join(root, "tools", "vdl2hip_iqfile")
    src = os.path.join(root, "tools", "vdl2hip_iqfile.c")
    if not os.path.exists(out_path) or os.path.getmtime(out_path) < max(os.path.getmtime(src), os.path.getmtime(LIB) if os.path.exists(LIB) else 0):
        subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-Wall", "-Wextra", "-I", os.path.join(root, "include"), src,
                               "-L", HERE, "-lvdl2hip", "-Wl,-rpath," + HERE, "-o", out_path])
    return out_path


if __name__ == "__main__":
    print(build(force=True, verbose=True))
