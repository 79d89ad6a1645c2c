"""Build libeda_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

The shared object lands next to the sources (eda_amd/csrc/libeda_hip.so) so it
travels with the repo snapshot to the GPU box; nothing is JIT-cached elsewhere.
"""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libeda_hip.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",        # every fused multiply-add in the kernels is explicit
    "-munsafe-fp-atomics",      # hardware fp32 atomic add for the scatter-add gradients
    "-Wall", "-Wno-unused-function",
]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [
        os.path.join(HERE, "..", "include", "eda_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every .hip source into one shared library.  Returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_variant(name, extra_flags, verbose=False):
    """An experiment build of the same sources with extra -D flags -> csrc/libeda_hip_<name>.so (select it with
    EDA_HIP_LIB=<path>; tools/mha2_phase_profile.py uses -DEDA_MHA2_PROFILE)."""
    out = os.path.join(CSRC, f"libeda_hip_{name}.so")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + list(extra_flags) + sources() + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
