"""Builds libpse_amd.so (HIP kernels + C-ABI) and the pybind11 host module in-tree with hipcc for gfx950.

Run as `python -m pse_amd.build` or through __graft_entry__.build(). No cmake: three translation units.
"""
import os
import subprocess
import sys
import sysconfig
import time

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


LIB_UNITS = (("pse_kernels.hip", HIPFLAGS), ("pse_farfield.hip", HIPFLAGS + FARFLAGS), ("pse_capi.hip", HIPFLAGS), ("pse_local.hip", HIPFLAGS), ("pse_zfft.hip", HIPFLAGS),
             ("pse_params.cpp", ["-x", "c++"]), ("pse_host_api.cpp", ["-x", "c++"]))
_report = []   # what build_all did, one entry per artefact: (name, "compiled" | "reused")


def build_lib(force=False):
    srcs = [x for x in _all_sources() if not x.endswith("asan_stub.cpp")]
    if not force and _newer(LIB, srcs):
        _report.append(("libpse_amd.so", "reused"))
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    objs = [os.path.join(CSRC, name.rsplit(".", 1)[0] + ".o") for name, _ in LIB_UNITS]

    def unit(k):   # a unit whose object is newer than every source is kept (the units share the headers: any header rebuilds all)
        name, extra = LIB_UNITS[k]
        deps = [os.path.join(CSRC, name)] + [x for x in srcs if x.endswith(".h") or x.endswith("build.py")]
        if force or not _newer(objs[k], deps):
            _run([HIPCC, *CXXFLAGS, *extra, "-c", os.path.join(CSRC, name), "-o", objs[k]])
    with ThreadPoolExecutor(max_workers=min(len(LIB_UNITS), os.cpu_count() or 1)) as pool:
        list(pool.map(unit, range(len(LIB_UNITS))))
    _run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs,
          f"-L{ROCM}/lib", "-lrocfft", "-lrccl", "-lpthread", "-ldl", f"-Wl,-rpath,{ROCM}/lib"])
    _report.append(("libpse_amd.so", "compiled"))
    return LIB


def _module_sources():
    host = os.path.join(CSRC, "host")
    srcs = [os.path.join(host, f) for f in sorted(os.listdir(host)) if f.endswith(".cc")]
    return srcs, srcs + [os.path.join(host, f) for f in os.listdir(host)]


def build_module(force=False):
    """pybind11 module _PSEv1: the C++ host side (Stokes, ShearFunction*, module) over the C-ABI."""
    import pybind11
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    target = os.path.join(HERE, "_PSEv1" + ext)
    if not os.path.isdir(os.path.join(CSRC, "host")):
        return None
    srcs, deps = _module_sources()
    if not force and _newer(target, deps + [LIB]):
        _report.append(("_PSEv1", "reused"))
        return target
    inc = ["-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(HERE, "..", "include")]
    _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", *inc, *srcs, "-o", target,
          f"-L{HERE}", "-lpse_amd", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{ROCM}/lib"])
    _report.append(("_PSEv1", "compiled"))
    return target


def build_all(force=None):
    """force=None: PSE_BUILD_FORCE=1 in the environment compiles everything again; otherwise artefacts newer than every source
    (the ones that travel with a gpurun push are) are reused.  Prints one line saying which it was."""
    if force is None:
        force = os.environ.get("PSE_BUILD_FORCE", "") not in ("", "0")
    del _report[:]
    t0 = time.time()
    build_lib(force)
    build_module(force)
    print("pse_amd.build: " + ", ".join(f"{n} {how}" for n, how in _report) + f" ({time.time() - t0:.1f} s"
          + (", forced" if force else "") + ")", flush=True)


# ---- CPU sanitizer build: never on the GPU, never the product --------------------------------------------
ASAN_DIR = os.path.join(HERE, "..", "build", "asan")
ASAN_MARKER = ".pse_asan_build"   # (also named in pse_amd/_lib.py, which must stay importable without this module's dependencies)
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def build_asan():
    """g++ -fsanitize=address,undefined builds of everything of the product that runs on the host: the parameter rule + table
    builder (pse_params.cpp), the host-only C-ABI incl. the Lanczos tridiagonal solver (pse_host_api.cpp), the C++ host classes
    and their pybind11 module (csrc/host/*, against csrc/asan_stub.cpp standing in for the device entry points).
    Output: build/asan/{libpse_amd.so,_PSEv1*.so}; tools/asan.py adds the checker and runs the CPU tests on them."""
    import pybind11
    out = os.path.abspath(ASAN_DIR)
    os.makedirs(out, exist_ok=True)
    inc = ["-I" + os.path.join(HERE, "..", "include")]
    lib = os.path.join(out, "libpse_amd.so")
    _run(["g++", "-std=c++17", "-fPIC", "-shared", "-Wall", *SAN, *inc, os.path.join(CSRC, "pse_params.cpp"),
          os.path.join(CSRC, "pse_host_api.cpp"), os.path.join(CSRC, "asan_stub.cpp"), "-o", lib])
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    srcs, _ = _module_sources()
    _run(["g++", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", *SAN, *inc, "-I" + pybind11.get_include(),
          "-I" + sysconfig.get_paths()["include"], *srcs, "-o", os.path.join(out, "_PSEv1" + ext), f"-L{out}", "-lpse_amd",
          "-Wl,-rpath,$ORIGIN"])
    with open(os.path.join(out, ASAN_MARKER), "w") as f:   # what makes the package accept the directory (pse_amd/_lib.py asan_dir)
        f.write("CPU sanitizer build of the host side of pse_amd: device entry points are stubs\n")
    print("pse_amd.build: sanitizer build in " + out, flush=True)
    return out


if __name__ == "__main__":
    if "--asan" in sys.argv:
        build_asan()
    else:
        build_all(force=True if "--force" in sys.argv else None)
