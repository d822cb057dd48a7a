#!/usr/bin/env python3
"""Pins the Lanczos DRIVER to the reference's own text (the last structural rule of the path that was pinned to mathematics only):
how gpu_stokes_BrealLanczos_wrap (PSEv1/Brownian.cu:357-765) grows the basis -- m_in - 1 iterations first, then one more per
pass of the while loop --, which scalars it keeps (alpha_j = v_j.(M v_j - beta_j v_{j-1}), beta_{j+1} = |v|), how it forms
T_m^{1/2} e_1 from the eigen-decomposition, its step norm sqrt(|u_m - u_{m-1}|^2 / (psi.M psi / |psi|^2)), its stopping and
breakdown rules (stepnorm <= cheb_error, m = m_max, |v| < 1e-8) and its final scaling |psi| sqrt(2 T / dt).

Runs in the build container only.  It READS the host function and the two helper kernels it launches
(gpu_stokes_LinearCombination_kernel, gpu_stokes_MatVecMultiply_kernel: PSEv1/Helper.cu:113-134,251-281) from /root/reference at
run time and executes that text with the C-subset interpreter tests/golden/cmini.py; the fixture
(tests/golden/reference_lanczos.json) holds inputs and results only.

What is not plain C in that text and is therefore supplied here:
  * `kernel<<<grid, threads, shmem>>>(args)`: the launch configuration is dropped and the callable runs every thread of the grid;
  * gpu_stokes_Mreal_kernel: the near-field operator, pinned separately (tests/test_reference_kernels.py::test_pair_formula) --
    here a dense symmetric positive definite matrix applied to the .xyz of the vector;
  * gpu_stokes_DotStepOne/Two_kernel (Helper.cu:146-243): a shared-memory tree reduction; the stand-in sums a.xyz . b.xyz over the
    group, and check_dot_kernels() runs the TEXT of both kernels once, with real barriers (one Python thread per GPU thread),
    against that stand-in before anything is written;
  * LAPACKE_spteqr (eigenvalues descending, eigenvectors in the columns of a row-major matrix): numpy.linalg.eigh;
  * cudaMalloc / cudaMemcpy / malloc / free on Python lists (sizeof counts elements).
Arithmetic is double precision throughout (the build's type; the reference's float scalars are its precision, not its rule).

  python3 tests/golden/make_lanczos_fixture.py       # rewrites tests/golden/reference_lanczos.json
"""
import json
import math
import os
import re
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from cmini import CError, Machine, Ptr, Ref, SizeOf, Vec, function_source, parameter_names   # noqa: E402

REF = os.environ.get("PSE_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "reference_lanczos.json")


def read(path):
    with open(os.path.join(REF, path)) as f:
        return f.read()


def make(tname):
    return lambda *a: Vec(tname, *a)


MATH = {"sqrtf": math.sqrt, "sqrt": math.sqrt, "make_scalar3": make("Scalar3"), "make_scalar4": make("Scalar4"),
        "dot": lambda a, b: sum(a.f[n] * b.f[n] for n in a.f)}


def base(p):
    """(list, offset) of a pointer argument."""
    return (p.arr, p.off) if isinstance(p, Ptr) else (p, 0)


class Kernel:
    """A __global__ function of Helper.cu run over a 1-D grid, one interpreter pass per thread (no barriers inside)."""

    def __init__(self, name):
        params, body = function_source(read("PSEv1/Helper.cu"), "void " + name)
        self.names = [n for _, n in parameter_names(params)]
        self.m = Machine(MATH)
        self.tree = self.m.parse(body)

    def __call__(self, *args):
        env = dict(zip(self.names, args))
        n = int(env["group_size"])
        for t in range(n + 3):                                  # a few threads past the end: the kernels guard themselves
            env.update(blockDim=Vec("dim3", 64, 1, 1), blockIdx=Vec("dim3", t // 64, 0, 0), threadIdx=Vec("dim3", t % 64, 0, 0))
            self.m.run(self.tree, env)


def dot_standin(state):
    def step_one(a, b, dot_sum, group_size, members):
        (aa, ao), (ba, bo) = base(a), base(b)
        s = 0.0
        for g in range(int(group_size)):
            i = members[g]
            s += sum(aa[ao + i].f[c] * ba[bo + i].f[c] for c in "xyz")
        for k in range(len(dot_sum)):
            dot_sum[k] = 0.0
        dot_sum[0] = s
        state["dots"] += 1

    def step_two(dot_sum, n):
        dot_sum[0] = sum(dot_sum[:int(n)])
    return step_one, step_two


def check_dot_kernels():
    """The text of gpu_stokes_DotStepOne_kernel and gpu_stokes_DotStepTwo_kernel, every thread of the grid a Python thread and
    __syncthreads a real barrier, against the stand-in."""
    text = read("PSEv1/Helper.cu")
    rng = np.random.default_rng(3)
    n, block = 37, 16                                           # two-and-a-bit blocks of a power-of-two size
    members = list(range(n))
    a = [Vec("Scalar4", *rng.normal(size=4)) for _ in range(n)]
    b = [Vec("Scalar4", *rng.normal(size=4)) for _ in range(n)]
    nblocks = n // block + 1
    dot_sum = [float("nan")] * nblocks

    def run_grid(name, grid, env):
        params, body = function_source(text, "void " + name)
        names = [q for _, q in parameter_names(params)]
        errors = []
        for blk in range(grid):
            shared = [0.0] * block
            barrier = threading.Barrier(block)

            def thread(t, blk=blk, shared=shared, barrier=barrier):
                try:
                    m = Machine(dict(MATH, __syncthreads=barrier.wait))
                    e = dict(zip(names, env), partial_sum=shared, blockDim=Vec("dim3", block, 1, 1), blockIdx=Vec("dim3", blk, 0, 0),
                             threadIdx=Vec("dim3", t, 0, 0))
                    m.run(m.parse(body), e)
                except Exception as ex:   # noqa: BLE001
                    errors.append(ex)
                    barrier.abort()
            ts = [threading.Thread(target=thread, args=(t,)) for t in range(block)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        if errors:
            raise errors[0]

    run_grid("gpu_stokes_DotStepOne_kernel", nblocks, [a, b, dot_sum, n, members])
    run_grid("gpu_stokes_DotStepTwo_kernel", 1, [dot_sum, nblocks])
    want = sum(a[i].f[c] * b[i].f[c] for i in range(n) for c in "xyz")
    assert abs(dot_sum[0] - want) < 1e-13 * max(1.0, abs(want)), (dot_sum[0], want)


def spteqr(layout, compz, n, d, e, z, ldz):
    (da, do), (ea, eo), (za, zo) = base(d), base(e), base(z)
    n = int(n)
    T = np.diag([da[do + i] for i in range(n)])
    for i in range(n - 1):
        T[i, i + 1] = T[i + 1, i] = ea[eo + i]
    w, V = np.linalg.eigh(T)
    w, V = w[::-1], V[:, ::-1]                                  # spteqr: eigenvalues in descending order
    for i in range(n):
        da[do + i] = float(w[i])
        for j in range(n):
            za[zo + i * int(ldz) + j] = float(V[i, j])          # row-major: column j is the eigenvector of eigenvalue j
    for i in range(n - 1):
        ea[eo + i] = 0.0
    return 0


def alloc(count):
    t = getattr(count, "tname", "Scalar")
    if t == "Scalar4":
        return [Vec("Scalar4", float("nan"), float("nan"), float("nan"), 0.0) for _ in range(int(count))]
    return [float("nan")] * int(count)


def cuda_malloc(ref, count):
    ref.put(alloc(count))
    return 0


def memcpy(dst, src, count, _kind):
    n = int(count)
    if isinstance(dst, Ref):                                    # cudaMemcpy(&scalar, d_array, sizeof(Scalar), ...)
        sa, so = base(src)
        dst.put(sa[so])
        return 0
    (da, do), (sa, so) = base(dst), base(src)
    for k in range(n):
        v = sa[so + k]
        da[do + k] = v.copy() if isinstance(v, Vec) else v
    return 0


class Exit(Exception):
    pass


def run_reference(M, psi, m_in, tol, T, dt):
    """Execute the reference's driver on the dense operator M (3n x 3n) and the vector psi (n x 3); returns (velocity, m, log)."""
    n = len(psi)
    _, body = function_source(read("PSEv1/Brownian.cu"), "void gpu_stokes_BrealLanczos_wrap")
    body = re.sub(r"<<<.*?>>>", "", body, flags=re.S)
    state = {"dots": 0, "matvecs": 0}

    def mreal(d_pos, d_out, d_in, group_size, *rest):
        x = np.array([[v.f[c] for c in "xyz"] for v in d_in]).ravel()
        y = (M @ x).reshape(n, 3)
        for i in range(n):
            d_out[i] = Vec("Scalar4", y[i, 0], y[i, 1], y[i, 2], d_out[i].f["w"] if isinstance(d_out[i], Vec) else 0.0)
        state["matvecs"] += 1

    def do_exit(code):
        raise Exit(code)

    one, two = dot_standin(state)
    builtins = dict(MATH, malloc=alloc, free=lambda p: None, cudaMalloc=cuda_malloc, cudaFree=lambda p: 0, cudaMemcpy=memcpy,
                    cudaMemcpyDeviceToDevice=0, cudaMemcpyDeviceToHost=1, cudaMemcpyHostToDevice=2, NULL=None,
                    LAPACKE_spteqr=spteqr, LAPACK_ROW_MAJOR=101, EXIT_FAILURE=1, printf=lambda *a: None, exit=do_exit,
                    gpu_stokes_Mreal_kernel=mreal, gpu_stokes_DotStepOne_kernel=one, gpu_stokes_DotStepTwo_kernel=two,
                    gpu_stokes_LinearCombination_kernel=Kernel("gpu_stokes_LinearCombination_kernel"),
                    gpu_stokes_MatVecMultiply_kernel=Kernel("gpu_stokes_MatVecMultiply_kernel"))
    mach = Machine(builtins)
    d_psi = [Vec("Scalar4", *row, 0.0) for row in psi]
    d_vel = [Vec("Scalar4", float("nan"), float("nan"), float("nan"), 0.0) for _ in range(n)]
    env = dict(d_psi=d_psi, d_pos=None, d_group_members=list(range(n)), group_size=n, box=None, dt=dt, d_vel=d_vel, T=T, timestep=0,
               seed=0, xi=0.5, ewald_cut=0.0, ewald_dr=0.0, ewald_n=0, d_ewaldC1=None, d_n_neigh=None, d_nlist=None, d_headlist=None,
               m=int(m_in), cheb_error=tol, grid=None, threads=None, gridBlockSize=0, gridNBlock=0, gridh=None, self=0.0)
    out = mach.run(mach.parse(body), env)
    vel = [[v.f[c] for c in "xyz"] for v in d_vel]
    return vel, int(out["m"]), state


def spd_matrix(n, seed, cond):
    """A dense symmetric positive definite 3n x 3n operator with eigenvalues spread over [1, cond]."""
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(3 * n, 3 * n)))
    lam = np.exp(rng.uniform(0.0, math.log(cond), 3 * n))
    return (Q * lam) @ Q.T


def main():
    check_dot_kernels()
    cases = []
    for name, n, seed, cond, m_in, tol, T, dt in (("m_in 2, tol 1e-3", 12, 1, 8.0, 2, 1e-3, 1.0, 1e-3),
                                                  ("m_in 5 (warm start), tol 1e-3", 12, 2, 8.0, 5, 1e-3, 0.7, 2e-3),
                                                  ("m_in 1", 10, 3, 4.0, 1, 1e-2, 1.0, 1e-3),
                                                  ("tol 1e-6", 14, 4, 20.0, 2, 1e-6, 1.0, 1e-3),
                                                  ("m_in beyond convergence", 8, 5, 3.0, 12, 1e-3, 2.0, 1e-3),
                                                  ("ill conditioned", 12, 6, 400.0, 2, 1e-4, 1.0, 5e-4)):
        M = spd_matrix(n, seed, cond)
        psi = np.random.default_rng(100 + seed).normal(size=(n, 3))
        vel, m, st = run_reference(M, psi, m_in, tol, T, dt)
        cases.append(dict(name=name, M=M.tolist(), psi=psi.tolist(), m_in=m_in, tol=tol, T=T, dt=dt, vel=vel, m=m, matvecs=st["matvecs"]))
        print(name, "-> m", m, "mat-vecs", st["matvecs"], flush=True)
    # breakdown: psi inside a 3-dimensional invariant subspace -- |v| < 1e-8 ends the iteration (Brownian.cu:503-506, 655-658)
    n = 8
    rng = np.random.default_rng(9)
    Q, _ = np.linalg.qr(rng.normal(size=(3 * n, 3 * n)))
    lam = np.full(3 * n, 2.0); lam[:3] = (1.0, 3.0, 7.0)
    M = (Q * lam) @ Q.T
    psi = (Q[:, :3] @ np.array([0.6, -1.1, 0.4])).reshape(n, 3)
    vel, m, st = run_reference(M, psi, 2, 1e-12, 1.0, 1e-3)
    cases.append(dict(name="breakdown in an invariant subspace", M=M.tolist(), psi=psi.tolist(), m_in=2, tol=1e-12, T=1.0, dt=1e-3, vel=vel,
                      m=m, matvecs=st["matvecs"]))
    print("breakdown -> m", m, "mat-vecs", st["matvecs"])
    with open(OUT, "w") as f:
        json.dump({"_source": "tests/golden/make_lanczos_fixture.py: PSEv1/Brownian.cu gpu_stokes_BrealLanczos_wrap executed by tests/golden/cmini.py",
                   "cases": cases}, f)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
