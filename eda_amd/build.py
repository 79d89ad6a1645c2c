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
    "-fvisibility=hidden",      # exports = the names include/eda_hip.h declares (its visibility pragma), nothing else
    "-Wall", "-Wno-unused-function",
]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "eda_hip.h")]


def _dep_time(depfile, fallback):
    """Newest mtime among the repo headers a previous compile recorded (-MD); every header's if there is no record."""
    try:
        words = open(depfile).read().replace("\\\n", " ").split()
    except OSError:
        return fallback
    root = os.path.abspath(os.path.join(HERE, ".."))
    t = 0.0
    for w in words[1:]:
        a = os.path.abspath(w)
        if a.startswith(root):
            try:
                t = max(t, os.path.getmtime(a))
            except OSError:
                return fallback
    return t


def _compile_objects(objdir, extra_flags, force, verbose):
    """One object per source (hipcc -c), in parallel; an object is rebuilt when its source or any header is newer."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags)
    hdr_t = max(os.path.getmtime(d) for d in _deps())
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        newest = max(os.path.getmtime(src), _dep_time(obj + ".d", hdr_t))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            jobs.append([hipcc] + flags + ["-MD", "-MF", obj + ".d", "-c", src, "-o", obj])
    if jobs:
        if verbose:
            for j in jobs:
                print(" ".join(j))
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(subprocess.check_call, jobs))
    return objs, bool(jobs)


def _link(objs, out, verbose):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC",
           "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")] + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)


def build(force=False, verbose=False):
    """Compile every .hip source (one object each, in parallel, only what is out of date) and link the shared
    library.  Returns its path."""
    objs, rebuilt = _compile_objects(os.path.join(CSRC, "_obj"), [], force, verbose)
    if rebuilt or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        _link(objs, LIB, verbose)
    return LIB


def build_variant(name, extra_flags, verbose=False):
    """An experiment build of the same sources with extra -D flags -> csrc/libeda_hip_<name>.so (select it with
    EDA_HIP_LIB=<path>; tools/mha2_phase_profile.py uses -DEDA_MHA2_PROFILE)."""
    out = os.path.join(CSRC, f"libeda_hip_{name}.so")
    objs, _ = _compile_objects(os.path.join(CSRC, "_obj_" + name), extra_flags, False, verbose)
    _link(objs, out, verbose)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
