"""Builds libpse_amd.so (HIP kernels + C-ABI) and the pybind11 host module in-tree with hipcc for gfx950.

Run as `python -m pse_amd.build` or through __graft_entry__.build(). No cmake: three translation units.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
LIB = os.path.join(HERE, "libpse_amd.so")
ARCH = "gfx950"

CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-Wno-unused-function"]
HIPFLAGS = [f"--offload-arch={ARCH}", "-munsafe-fp-atomics", "-ffp-contract=fast"]
# the spread kernel keeps its TZ x 3 accumulators in registers across a switch over the particle's z offset: SimplifyCFG's
# sinking of the cases' common stores turns the accumulators into pointer phis, which leaves them in scratch memory
FARFLAGS = ["-mllvm", "-simplifycfg-sink-common=false"]


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0 or os.environ.get("PSE_BUILD_VERBOSE"):
        print(" ".join(cmd), flush=True)
        print(r.stdout, flush=True)
    if r.returncode != 0:
        raise RuntimeError("build step failed: " + " ".join(cmd))


def _all_sources():
    out = []
    for root, _, files in os.walk(CSRC):
        out += [os.path.join(root, f) for f in files if f.endswith((".hip", ".h", ".cpp", ".cc"))]
    out.append(os.path.join(HERE, "..", "include", "pse_amd.h"))
    out.append(os.path.abspath(__file__))
    return out


def build_lib(force=False):
    srcs = _all_sources()
    if not force and _newer(LIB, srcs):
        return LIB
    objs = []
    for name, extra in (("pse_kernels.hip", HIPFLAGS), ("pse_farfield.hip", HIPFLAGS + FARFLAGS), ("pse_capi.hip", HIPFLAGS), ("pse_params.cpp", ["-x", "c++"])):
        obj = os.path.join(CSRC, name.rsplit(".", 1)[0] + ".o")
        _run([HIPCC, *CXXFLAGS, *extra, "-c", os.path.join(CSRC, name), "-o", obj])
        objs.append(obj)
    _run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs,
          f"-L{ROCM}/lib", "-lrocfft", "-lrccl", "-lpthread", "-ldl", f"-Wl,-rpath,{ROCM}/lib"])
    return LIB


def build_module(force=False):
    """pybind11 module _PSEv1: the C++ host side (Stokes, ShearFunction*, module) over the C-ABI."""
    import pybind11
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    target = os.path.join(HERE, "_PSEv1" + ext)
    host = os.path.join(CSRC, "host")
    if not os.path.isdir(host):
        return None
    srcs = [os.path.join(host, f) for f in sorted(os.listdir(host)) if f.endswith(".cc")]
    deps = srcs + [os.path.join(host, f) for f in os.listdir(host)] + [LIB]
    if not force and _newer(target, deps):
        return target
    inc = ["-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(HERE, "..", "include")]
    _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", *inc, *srcs, "-o", target,
          f"-L{HERE}", "-lpse_amd", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{ROCM}/lib"])
    return target


def build_all(force=False):
    build_lib(force)
    build_module(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
