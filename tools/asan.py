#!/usr/bin/env python3
"""CPU sanitizer run.  GPU AddressSanitizer is not available on the MI355X pool, so everything on the path that
runs on the HOST is built with gcc -fsanitize=address,undefined and the CPU test suite is run against it:

  product   pse_params.cpp (parameter rule, real-space table by quadrature + Chebyshev fits), pse_host_api.cpp (host-only C-ABI,
            Lanczos tridiagonal solver), csrc/host/* (Stokes, ShearFunction*, pybind11 module) over csrc/asan_stub.cpp -- built by
            `python -m pse_amd.build --asan` into build/asan/
  checker   oracle/pse_oracle.c -- built here into build/asan/libpse_oracle.so

  python tools/asan.py            build both, run `pytest tests -m "not gpu"` with PSE_ASAN_DIR set and the sanitizer runtimes preloaded
  python tools/asan.py -k oracle  extra arguments go to pytest
Exit code: pytest's; 99 if a sanitizer reported.  tests/test_abi.py::test_sanitizer_build_is_in_use asserts that the run really
loaded the instrumented libraries."""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    from pse_amd import build as b
    out = b.build_asan()
    subprocess.check_call(["gcc", "-std=gnu11", "-fPIC", "-shared", "-fopenmp", "-Wall", *b.SAN, os.path.join(ROOT, "oracle", "pse_oracle.c"),
                           "-o", os.path.join(out, "libpse_oracle.so"), "-lm"])
    rt = [subprocess.run(["gcc", "-print-file-name=" + n], stdout=subprocess.PIPE, text=True).stdout.strip() for n in ("libasan.so", "libubsan.so")]
    env = dict(os.environ, PSE_ASAN_DIR=out, LD_PRELOAD=":".join(rt),
               # the interpreter is not instrumented: its arena allocator would be reported as leaking at exit
               ASAN_OPTIONS="detect_leaks=0:exitcode=99:allocator_may_return_null=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


if __name__ == "__main__":
    raise SystemExit(main())
