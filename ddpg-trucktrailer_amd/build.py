"""Build libttenv.so (HIP, gfx950) in-tree with hipcc.  No JIT cache: the .so sits next to this
file so that it travels to the GPU box with the repo snapshot."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", f) for f in ("ttenv.hip", "ttnet.hip", "ttnet_split.hip", "ttlearn.hip", "ttp2p.hip")]
HDR = [os.path.join(ROOT, "include", "ttenv.h"), os.path.join(HERE, "csrc", "ttnet_common.h"), os.path.join(HERE, "csrc", "ttnet_pack.h"), os.path.join(HERE, "csrc", "ttp2p.h"), os.path.join(HERE, "csrc", "ttstamps.h")]
LIB = os.path.join(HERE, "libttenv.so")


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SRC + HDR)


def build(force=False, verbose=False, defines=(), out=None):
    """Compile every HIP source for gfx950 into ddpg-trucktrailer_amd/libttenv.so.
    `defines`/`out` build a kernel variant next to it (A/B timing via TT_LIB_PATH)."""
    if out is None and not force and not stale():
        return LIB
    out = out or LIB
    # kernarg preload: the first 16 dwords of a kernel's leading scalar / pointer arguments reach the wave in SGPRs instead of through
    # its first s_load round trip (the hot kernels put the pointers of their first loads there: csrc/ttlearn.hip, KernargWarm)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-kernarg-preload-count=16",
           "-I" + os.path.join(ROOT, "include"), "-o", out] + ["-D" + d for d in defines] + SRC
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


# diagnostic builds of the same sources (A/B timing and stamps through TT_LIB_PATH; tools/*.py, tools/profile_round*.sh)
VARIANTS = {
    "stamps": (("TT_STAMPS",), os.path.join(ROOT, "tools", "dbg", "libttenv_stamps.so")),    # wall-clock stamps per workgroup / phase
}


def build_variant(name, verbose=False):
    defines, out = VARIANTS[name]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if os.path.exists(out) and all(os.path.getmtime(p) <= os.path.getmtime(out) for p in SRC + HDR):
        return out
    return build(force=True, verbose=verbose, defines=defines, out=out)


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":
        print(build_variant(sys.argv[2], verbose=True))
    else:
        print(build(force=True, verbose=True))
