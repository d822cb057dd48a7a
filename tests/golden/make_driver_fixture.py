#!/usr/bin/env python3
"""Pins the two pieces of the reference's step that the kernel fixture left to restatement:

  * the HOST wrapper gpu_stokes_CombinedMobilityBrownian_wrap (PSEv1/Brownian.cu:772-923): which kernels run in which order for
    T = 0 and T > 0, and the numbers the host computes for them -- prefac = (2 xi^2 / pi / eta)^{3/2}, expfac = 2 xi^2 / eta, the
    gather weight quadW prefac with quadW = hx hy hz, the block edge B = min(P, 10), the (1, 1) coefficients that add the parts.
    The text is executed by tests/golden/cmini.py with every kernel / cuFFT / cudaMalloc call replaced by a recorder;
  * gpu_stokes_Contract_kernel (PSEv1/Mobility.cu:325-477; K8, the gather): support centring and wrap, sheared node position,
    weight, and its shared-memory tree reduction -- the text runs one Python thread per GPU thread with a real barrier for
    __syncthreads (blocks of 4^3 ... 10^3 threads, the latter with two support nodes per thread along each axis: P = 13);
  * gpu_stokes_BrownianGenerate_kernel (PSEv1/Brownian.cu:99-130; K14): the key of a particle's random numbers, their interval,
    the components written.

Runs in the build container only; reads /root/reference at run time; the fixture (tests/golden/reference_driver.json.gz) holds inputs
and results only.  HOOMD's BoxDim, texFetchScalar4 and make_scalar* are restated as in make_kernel_fixture.py.

  python3 tests/golden/make_driver_fixture.py
"""
import gzip
import json
import math
import os
import re
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from cmini import Machine, Ref, Vec, function_source, parameter_names   # noqa: E402
from make_kernel_fixture import BUILTINS, BoxDim, read                 # noqa: E402

OUT = os.path.join(HERE, "reference_driver.json.gz")


# ---------------------------------------------------------------------------------------------- the host wrapper
def run_wrapper(xi, eta, P, gridh, T):
    params, body = function_source(read("PSEv1/Brownian.cu"), "void gpu_stokes_CombinedMobilityBrownian_wrap")
    names = [n for _, n in parameter_names(params)]
    body = re.sub(r"<<<.*?>>>", "", body, flags=re.S)
    log = []

    def recorder(name, keep):
        def call(*args):
            log.append([name] + [float(args[i]) if isinstance(args[i], (int, float)) else str(args[i]) for i in keep])
        return call

    # argument positions of the scalars the host computes (counted in the reference's own parameter lists)
    def positions(fname, path, wanted):
        p, _ = function_source(read(path), "void " + fname)
        order = [n for _, n in parameter_names(p)]
        return [order.index(w) for w in wanted]

    builtins = dict(BUILTINS, dim3=lambda *a: Vec("dim3", *a), cudaMalloc=lambda ref, n: ref.put([None] * int(n)) or 0, cudaFree=lambda p: 0,
                    CUFFT_FORWARD="forward", CUFFT_INVERSE="inverse",
                    cufftExecC2C=recorder("cufftExecC2C", [3]),
                    gpu_stokes_BrownianGenerate_kernel=recorder("BrownianGenerate", []),
                    gpu_stokes_ZeroGrid_kernel=recorder("ZeroGrid", []),
                    gpu_stokes_Spread_kernel=recorder("Spread", positions("gpu_stokes_Spread_kernel", "PSEv1/Mobility.cu", ["P", "prefac", "expfac"])),
                    gpu_stokes_Green_kernel=recorder("Green", []),
                    gpu_stokes_BrownianGridGenerate_kernel=recorder("BrownianGridGenerate", positions("gpu_stokes_BrownianGridGenerate_kernel", "PSEv1/Brownian.cu", ["T", "dt", "quadW"])),
                    gpu_stokes_Contract_kernel=recorder("Contract", positions("gpu_stokes_Contract_kernel", "PSEv1/Mobility.cu", ["P", "prefac", "expfac"])),
                    gpu_stokes_Mreal_kernel=recorder("Mreal", []),
                    gpu_stokes_LinearCombination_kernel=recorder("LinearCombination", [3, 4]),
                    gpu_stokes_BrealLanczos_wrap=recorder("BrealLanczos", positions("gpu_stokes_BrealLanczos_wrap", "PSEv1/Brownian.cu", ["dt", "T", "cheb_error"])))
    m = Machine(builtins)
    env = {n: None for n in names}
    env.update(group_size=7, dt=1e-3, T=T, timestep=3, seed=5, xi=xi, eta=eta, P=P, Nx=8, Ny=8, Nz=8, m_Lanczos=2, N_total=7, NxNyNz=512,
               gridBlockSize=256, gridNBlock=2, gridh=Vec("Scalar3", *gridh), cheb_error=1e-3, self=0.3, ewald_cut=5.0, ewald_dr=1e-3, ewald_n=5000)
    m.run(m.parse(body), env)
    return log


# ---------------------------------------------------------------------------------------------- the gather kernel (K8)
def run_contract(pos, ugrid, grid, box, P, xi, eta):
    """d_vel of every particle from gpu_stokes_Contract_kernel's text: one block per particle, B^3 threads, real barriers."""
    _, body = function_source(read("PSEv1/Mobility.cu"), "void gpu_stokes_Contract_kernel")
    body = re.sub(r"extern\s+__shared__\s+Scalar3\s+shared\s*\[\s*\]\s*;", "", body)        # handed in by the launcher below
    Nx, Ny, Nz = grid
    B = min(P, 10)                                                  # Brownian.cu:822-824, pinned by the wrapper section
    c = 2.0 * xi * xi / eta
    gridh = (box[0] / Nx, box[1] / Ny, box[2] / Nz)
    prefac = gridh[0] * gridh[1] * gridh[2] * (c / math.pi) ** 1.5  # quadW * prefac, ditto
    g = [[Vec("Scalar2", float(v), 0.0) for v in ugrid[cmp].ravel()] for cmp in range(3)]
    d_pos = [Vec("Scalar4", *p, 0.0) for p in pos]
    d_vel = [Vec("Scalar4", float("nan"), float("nan"), float("nan"), 7.0) for _ in pos]
    members = list(range(len(pos)))
    tree_src = body
    for p in range(len(pos)):
        shared = [Vec("Scalar3", float("nan"), float("nan"), float("nan")) for _ in range(B * B * B + 1)]
        barrier = threading.Barrier(B * B * B)
        errors = []

        def thread(tx, ty, tz, p=p, shared=shared, barrier=barrier):
            try:
                m = Machine(dict(BUILTINS, __syncthreads=barrier.wait))
                m.run(m.parse(tree_src), dict(d_pos=d_pos, d_vel=d_vel, gridX=g[0], gridY=g[1], gridZ=g[2], group_size=len(pos), Nx=Nx, Ny=Ny, Nz=Nz,
                                              xi=xi, eta=eta, d_group_members=members, box=BoxDim(*box), P=P, gridh=Vec("Scalar3", *gridh),
                                              prefac=prefac, expfac=c, shared=shared, blockDim=Vec("dim3", B, B, B),
                                              blockIdx=Vec("dim3", p, 0, 0), threadIdx=Vec("dim3", tx, ty, tz)))
            except Exception as ex:   # noqa: BLE001
                errors.append(ex)
                barrier.abort()
        ts = [threading.Thread(target=thread, args=(tx, ty, tz)) for tx in range(B) for ty in range(B) for tz in range(B)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errors:
            raise errors[0]
    return [[v.f[k] for k in "xyzw"] for v in d_vel]


# ---------------------------------------------------------------------------------------------- the particle noise kernel (K14)
def run_psi(n_total, members, seed, timestep):
    """gpu_stokes_BrownianGenerate_kernel (PSEv1/Brownian.cu:99-130) from its text.  Saru is not in the tree: the stand-in hands
    particle idx the uniforms the port's Philox stream assigns to that particle, so what is pinned is what the kernel DOES with them
    -- the key (GLOBAL particle index, timestep + seed), the interval, which components are written."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import pse_port
    _, body = function_source(read("PSEv1/Brownian.cu"), "void gpu_stokes_BrownianGenerate_kernel")
    keys = []

    class Saru:
        def __init__(self, idx, key):
            keys.append((int(idx), int(key)))
            r = pse_port.philox4x32(np.uint64(idx) & np.uint64(0xFFFFFFFF), np.uint64(idx) >> np.uint64(32), timestep, pse_port.DOMAIN_PARTICLE,
                                    seed, pse_port.KEY1)
            self.u = [(float(x) + 0.5) * 2.0 ** -32 for x in r]
            self.k = 0

        def f(self, a, b):
            self.k += 1
            return a + (b - a) * self.u[self.k - 1]

    m = Machine(dict(BUILTINS, **{"detail::Saru": Saru}), extra_types=("detail::Saru",))
    tree = m.parse(body)
    d_psi = [Vec("Scalar4", -9.0, -9.0, -9.0, 4.5) for _ in range(n_total)]
    for t in range(len(members) + 5):
        m.run(tree, dict(d_psi=d_psi, group_size=len(members), d_group_members=members, timestep=timestep, seed=seed,
                         blockDim=Vec("dim3", 32, 1, 1), blockIdx=Vec("dim3", t // 32, 0, 0), threadIdx=Vec("dim3", t % 32, 0, 0)))
    assert sorted(keys) == sorted((i, timestep + seed) for i in members)
    return [[v.f[k] for k in "xyzw"] for v in d_psi]


def main():
    threading.stack_size(512 * 1024)
    out = {"_source": "tests/golden/make_driver_fixture.py: PSEv1/Brownian.cu gpu_stokes_CombinedMobilityBrownian_wrap and PSEv1/Mobility.cu "
                      "gpu_stokes_Contract_kernel executed by tests/golden/cmini.py", "wrapper": [], "contract": [], "psi": None}
    for xi, eta, P, gridh, T in ((0.5, 0.47, 6, (0.8, 0.9, 1.0), 1.0), (0.5, 0.47, 6, (0.8, 0.9, 1.0), 0.0), (0.31, 0.72, 13, (1.1, 1.1, 1.3), 0.5)):
        out["wrapper"].append(dict(xi=xi, eta=eta, P=P, gridh=gridh, T=T, calls=run_wrapper(xi, eta, P, gridh, T)))
    rng = np.random.default_rng(77)
    for grid, box, P, xi, eta, n in (((8, 9, 10), (7.0, 8.1, 9.5, 0.0), 4, 0.5, 0.6, 3), ((9, 8, 10), (8.0, 7.3, 9.1, 0.3), 5, 0.5, 0.5, 3),
                                     ((12, 10, 9), (10.0, 9.0, 8.4, -0.4), 6, 0.45, 0.55, 2), ((14, 13, 15), (13.0, 12.5, 14.0, 0.25), 13, 0.4, 0.3, 1)):
        f = rng.uniform(-0.5, 0.5, (n, 3))
        pos = np.empty((n, 3))
        pos[:, 1] = f[:, 1] * box[1]; pos[:, 2] = f[:, 2] * box[2]; pos[:, 0] = f[:, 0] * box[0] + box[3] * pos[:, 1]
        pos[0] = (-0.49 * box[0] + box[3] * pos[0, 1], pos[0, 1], 0.49 * box[2])        # supports that wrap at both ends
        ug = rng.normal(size=(3,) + grid)
        vel = run_contract(pos, ug, grid, box, P, xi, eta)
        out["contract"].append(dict(grid=grid, box=box, P=P, xi=xi, eta=eta, pos=pos.tolist(), ugrid=ug.tolist(), vel=vel))
        print("contract", grid, "P", P, "done", flush=True)
    members = [0, 3, 4, 9, 17, 18, 40]
    out["psi"] = dict(n_total=41, members=members, seed=987654, timestep=12, psi=run_psi(41, members, 987654, 12))
    with gzip.GzipFile(OUT, "wb", mtime=0) as f:
        f.write(json.dumps(out).encode())
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
