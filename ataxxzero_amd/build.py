"""Build libataxxzero_hip.so in-tree with hipcc for gfx950 (MI355X).

hipcc cross-compiles without a GPU, so this runs on the CPU-only build box; the
built .so travels to the GPU box with the repo snapshot.  No cmake, no torch
extension machinery: five translation units, one link.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libataxxzero_hip.so")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function",
          "-I" + os.path.join(os.path.dirname(HERE), "include")]
# Tree kernels carry the bit-exact f32 contract with the oracle: no fused contraction.
UNITS = [
    ("engine.hip", ["-ffp-contract=off"]),
    ("rules_api.hip", ["-ffp-contract=off"]),
    ("net_kernels.hip", []),
    ("common.cpp", []),
    ("json.cpp", []),
    ("ref_abi.cpp", []),
]


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
        [os.path.join(os.path.dirname(HERE), "include", "ataxxzero_hip.h"), os.path.abspath(__file__)]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=False):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(LIB):
            return LIB  # prebuilt library shipped with the snapshot
        raise RuntimeError("hipcc not found and no prebuilt %s" % LIB)
    deps = _deps()
    sources = [os.path.join(CSRC, src) for src, _ in UNITS]
    if not force and not _stale(LIB, sources + deps):
        return LIB  # up to date (the object directory does not travel with the snapshot; the library does)
    os.makedirs(OBJ, exist_ok=True)
    objs, procs = [], []
    for src, extra in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + deps):
            cmd = [hipcc] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = []
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append("%s:\n%s" % (src, out.decode(errors="replace")))
        elif verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(failed))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        subprocess.check_call(cmd)
    return LIB


def build_variant(name, extra_flags, force=False, replace=None):
    """A second copy of the library compiled with extra flags (e.g. ["-DAZH_OOBZERO=0"]), for A/B runs and for the test
    that keeps the tower's fallback build alive.  Load it with AZH_LIB=<path> in a fresh process.  Objects and library
    live under csrc/_obj/variants/<name>/ (never shipped).  `replace` = {"engine.hip": "/path/to/another/engine.hip"}
    compiles a unit from another file (an earlier version of a kernel against the current one, same box, same call)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: a variant build needs the compiler")
    out_dir = os.path.join(OBJ, "variants", name)
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libataxxzero_hip.so")
    deps = _deps()
    objs, procs = [], []
    for src, extra in UNITS:
        s = (replace or {}).get(src) or os.path.join(CSRC, src)
        o = os.path.join(out_dir, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + deps):
            procs.append((src, subprocess.Popen([hipcc] + COMMON + extra + list(extra_flags) + ["-I" + CSRC, "-c", s, "-o", o],
                                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = ["%s:\n%s" % (src, out.decode(errors="replace")) for src, p in procs
              for out in [p.communicate()[0]] if p.returncode != 0]
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(failed))
    if force or _stale(lib, objs):
        subprocess.check_call([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib] + objs)
    return lib


# Bench-only target: the reference's architecture (one host thread per game + batched evaluator) as the CPU
# baseline bench.py times beside the GPU path.  A separate library on purpose: the product library has no CPU path.
BASELINE_SRC = os.path.join(os.path.dirname(HERE), "tools", "cpu_baseline", "host_selfplay.hip")
BASELINE_LIB = os.path.join(os.path.dirname(HERE), "tools", "cpu_baseline", "libazh_cpu_baseline.so")


def build_cpu_baseline(force=False):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(BASELINE_LIB):
            return BASELINE_LIB
        raise RuntimeError("hipcc not found and no prebuilt %s" % BASELINE_LIB)
    if force or _stale(BASELINE_LIB, [BASELINE_SRC, os.path.join(CSRC, "azh_device.h")]):
        subprocess.check_call([hipcc, "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=" + ARCH, "-pthread",
                               BASELINE_SRC, "-o", BASELINE_LIB])
    return BASELINE_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_cpu_baseline(force="--force" in sys.argv))
