#!/usr/bin/env python3
"""Pins the STRUCTURAL rules of the reference's kernels: which nodes a particle's support covers and how
they wrap, the sheared node position and weight, the FFT index folding of the wave vectors, the Green projector, the Hermitian
half-space / Nyquist / conjugate bookkeeping of the k-space noise, the pair formula with its table lookup, the Euler step.

Runs in the build container only.  It READS the text of the reference's kernels from /root/reference at run time,

  PSEv1/Mobility.cu  gpu_stokes_Spread_kernel (:114-252), gpu_stokes_Green_kernel (:264-299), gpu_stokes_Mreal_kernel (:594-687)
  PSEv1/Helper.cu    gpu_stokes_SetGridk_kernel (:285-332)
  PSEv1/Brownian.cu  gpu_stokes_BrownianGridGenerate_kernel (:153-345)
  PSEv1/Stokes.cu    gpu_stokes_step_one_kernel (:137-192)
  PSEv1/Stokes.cc    the closed forms that fill the real-space table (:348-406)

executes it thread by thread with the C-subset interpreter tests/golden/cmini.py on small inputs, and writes inputs and
results to tests/golden/reference_kernels.json.gz.  The fixture is data; no reference text is stored and none travels.

What is NOT in the reference tree and therefore restated here: HOOMD's BoxDim (getL, getTiltFactorXY,
makeFraction, minImage, wrap -- box centred on the origin, y images shift x by xy Ly), texFetchScalar4 (a plain load),
atomicAdd, make_scalar*, dot, __scalar2int_rd, and the Saru generator -- whose stream cannot be pinned: the stand-in hands each
thread the uniforms the port's Philox stream assigns to that thread's node, so that what is compared is everything the
kernel DOES with its random numbers.

Constants: the kernels are executed "as written" (pi = 3.1416926536 in the wave vectors, 3.1415926536 in the scale factor)
and with those two literals replaced by exact pi ("exact_pi": the deliberate difference of SURVEY.md 2.4); expf/sinf/sqrtf are
evaluated in double precision (the build's arithmetic type).

  python3 tests/golden/make_kernel_fixture.py       # rewrites tests/golden/reference_kernels.json.gz
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from cmini import Machine, Vec, function_source   # noqa: E402

REF = os.environ.get("PSE_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "reference_kernels.json.gz")
PI_K, PI_W = "3.1416926536", "3.1415926536"        # the two literals of PSEv1/Helper.cu:313-315,326


def read(path):
    with open(os.path.join(REF, path)) as f:
        return f.read()


# ---------------------------------------------------------------------------------------------- HOOMD pieces, restated
class BoxDim:
    """HOOMD-blue 2.x BoxDim for an xy-tilted cell centred on the origin (lo = -L/2), as the kernels use it."""

    def __init__(self, Lx, Ly, Lz, xy):
        self.Lx, self.Ly, self.Lz, self.xy = Lx, Ly, Lz, xy

    def getL(self):
        return Vec("Scalar3", self.Lx, self.Ly, self.Lz)

    def getTiltFactorXY(self):
        return self.xy

    def makeFraction(self, v):
        return Vec("Scalar3", (v.f["x"] - self.xy * v.f["y"]) / self.Lx + 0.5, v.f["y"] / self.Ly + 0.5, v.f["z"] / self.Lz + 0.5)

    def minImage(self, v):
        x, y, z = v.f["x"], v.f["y"], v.f["z"]
        z -= self.Lz * round(z / self.Lz)
        n = round(y / self.Ly)
        y -= self.Ly * n
        x -= self.Ly * self.xy * n
        x -= self.Lx * round(x / self.Lx)
        return Vec("Scalar3", x, y, z)

    def wrap(self, pos, image):
        n = math.floor(pos.f["z"] / self.Lz + 0.5)
        pos.f["z"] -= n * self.Lz; image.f["z"] += int(n)
        n = math.floor(pos.f["y"] / self.Ly + 0.5)
        pos.f["y"] -= n * self.Ly; pos.f["x"] -= n * self.xy * self.Ly; image.f["y"] += int(n)
        n = math.floor((pos.f["x"] - self.xy * pos.f["y"]) / self.Lx + 0.5)
        pos.f["x"] -= n * self.Lx; image.f["x"] += int(n)


def atomic_add(target, v):
    target.put(target.get() + v)


def make(tname):
    return lambda *a: Vec(tname, *a)


BUILTINS = {
    "expf": math.exp, "sqrtf": math.sqrt, "sinf": math.sin, "exp": math.exp, "sqrt": math.sqrt, "pow": math.pow, "erfc": math.erfc,
    "make_scalar2": make("Scalar2"), "make_scalar3": make("Scalar3"), "make_scalar4": make("Scalar4"),
    "dot": lambda a, b: sum(a.f[n] * b.f[n] for n in a.f),
    "__scalar2int_rd": lambda x: int(math.floor(x)),
    "texFetchScalar4": lambda arr, _tex, i: arr[i].copy(),
    "atomicAdd": atomic_add, "__syncthreads": lambda: None,
    "pos_tex": None, "tables1_tex": None,
}


def thread_1d(tid, block=256):
    return {"blockDim": Vec("dim3", block, 1, 1), "blockIdx": Vec("dim3", tid // block, 0, 0), "threadIdx": Vec("dim3", tid % block, 0, 0)}


def exact_pi(text):
    assert PI_K in text or PI_W in text
    return text.replace(PI_K, repr(math.pi)).replace(PI_W, repr(math.pi))


# ---------------------------------------------------------------------------------------------- wave vectors (K1)
def run_setgridk(grid, box, xi, eta, exact):
    body = function_source(read("PSEv1/Helper.cu"), "gpu_stokes_SetGridk_kernel")[1]
    m = Machine(BUILTINS)
    tree = m.parse(exact_pi(body) if exact else body)
    Nx, Ny, Nz = grid
    gridk = [Vec("Scalar4") for _ in range(Nx * Ny * Nz)]
    for tid in range(Nx * Ny * Nz):
        m.run(tree, dict(thread_1d(tid), gridk=gridk, Nx=Nx, Ny=Ny, Nz=Nz, NxNyNz=Nx * Ny * Nz, box=BoxDim(*box), xi=xi, eta=eta))
    return gridk


def gridk_array(gridk, grid):
    return np.array([[v.f[c] for c in "xyzw"] for v in gridk]).reshape(*grid, 4)


# ---------------------------------------------------------------------------------------------- Green (K5)
def run_green(gridk, f, grid):
    body = function_source(read("PSEv1/Mobility.cu"), "gpu_stokes_Green_kernel")[1]
    m = Machine(BUILTINS)
    tree = m.parse(body)
    ng = grid[0] * grid[1] * grid[2]
    g = [[Vec("Scalar2", f[c].flat[i].real, f[c].flat[i].imag) for i in range(ng)] for c in range(3)]
    for tid in range(ng):
        m.run(tree, dict(thread_1d(tid), gridX=g[0], gridY=g[1], gridZ=g[2], gridk=gridk, NxNyNz=ng))
    return np.array([[v.f["x"] + 1j * v.f["y"] for v in g[c]] for c in range(3)]).reshape(3, *grid)


# ---------------------------------------------------------------------------------------------- k-space noise (K6)
class SaruStandIn:
    """The port's uniforms for the node this thread owns, in the order the kernel draws them (reX, reY, reZ, imX, imY, imZ)."""
    table = None

    def __init__(self, idx, _key):
        self.vals = list(SaruStandIn.table[idx])

    def f(self, lo, hi):
        v = self.vals.pop(0)
        assert lo - 1e-12 <= v <= hi + 1e-12
        return v


def port_uniforms(grid, seed, timestep):
    """For every node of the full grid the (re[3], im[3]) the port's Philox stream gives it: the draw of a conjugate pair is keyed by
    the pair's canonical node; the partner holds the conjugate (oracle/pse_port.py noise_k).  Full-grid version of that rule."""
    from oracle import pse_port as pp
    Nx, Ny, Nz = grid
    i, j, k = np.meshgrid(np.arange(Nx), np.arange(Ny), np.arange(Nz), indexing="ij")
    own = (i * Ny + j) * Nz + k
    ic, jc, kc = (Nx - i) % Nx, (Ny - j) % Ny, (Nz - k) % Nz
    partner = (ic * Ny + jc) * Nz + kc
    # the port stores kz <= Nz/2 only: a node with kz > Nz/2 is the conjugate of its partner (which is stored); on the planes kz = 0
    # and kz = Nz/2 (even Nz) both are stored and the canonical one is the smaller linear index
    stored = k <= Nz // 2
    on_plane = (k == 0) | ((Nz % 2 == 0) & (k == Nz // 2))
    canon = np.where(stored & ~on_plane, own, np.where(stored, np.minimum(own, partner), partner))
    flip = canon != own
    cu = canon.astype(np.uint64)
    a = pp.philox4x32(cu & np.uint64(0xFFFFFFFF), cu >> np.uint64(32), timestep, pp.DOMAIN_GRID_A, seed, pp.KEY1)
    b = pp.philox4x32(cu & np.uint64(0xFFFFFFFF), cu >> np.uint64(32), timestep, pp.DOMAIN_GRID_B, seed, pp.KEY1)
    s = math.sqrt(1.5)
    re = np.stack([pp._u(a[0], s), pp._u(a[1], s), pp._u(a[2], s)], axis=-1)
    im = np.stack([pp._u(a[3], s), pp._u(b[0], s), pp._u(b[1], s)], axis=-1)
    im = np.where(flip[..., None], -im, im)
    return np.concatenate([re, im], axis=-1).reshape(-1, 6)


def run_brownian_grid(gridk, grid, kT, dt, h3, seed, timestep):
    body = function_source(read("PSEv1/Brownian.cu"), "gpu_stokes_BrownianGridGenerate_kernel")[1]
    SaruStandIn.table = port_uniforms(grid, seed, timestep)
    m = Machine(dict(BUILTINS, **{"detail::Saru": SaruStandIn}), extra_types=["detail::Saru"])
    tree = m.parse(body)
    Nx, Ny, Nz = grid
    ng = Nx * Ny * Nz
    g = [[Vec("Scalar2", 0.0, 0.0) for _ in range(ng)] for _ in range(3)]
    for tid in range(ng):
        m.run(tree, dict(thread_1d(tid), gridX=g[0], gridY=g[1], gridZ=g[2], gridk=gridk, NxNyNz=ng, Nx=Nx, Ny=Ny, Nz=Nz,
                         timestep=timestep, seed=seed, T=kT, dt=dt, quadW=h3))
    return np.array([[v.f["x"] + 1j * v.f["y"] for v in g[c]] for c in range(3)]).reshape(3, *grid)


# ---------------------------------------------------------------------------------------------- spread (K3)
def run_spread(pos, force, grid, box, P, xi, eta):
    body = function_source(read("PSEv1/Mobility.cu"), "gpu_stokes_Spread_kernel")[1]
    m = Machine(BUILTINS)
    tree = m.parse(body)
    Nx, Ny, Nz = grid
    ng = Nx * Ny * Nz
    g = [[Vec("Scalar2", 0.0, 0.0) for _ in range(ng)] for _ in range(3)]
    B = min(P, 10)                                             # PSEv1/Brownian.cu:823
    c = 2.0 * xi * xi / eta
    prefac, expfac = (c / math.pi) ** 1.5, c                   # PSEv1/Brownian.cu:828-829 (the values the host passes in)
    d_pos = [Vec("Scalar4", *p, 0.0) for p in pos]
    d_f = [Vec("Scalar4", *f, 0.0) for f in force]
    members = list(range(len(pos)))
    gridh = Vec("Scalar3", box[0] / Nx, box[1] / Ny, box[2] / Nz)
    for p in range(len(pos)):
        shared = [Vec("Scalar3"), Vec("Scalar3")]              # the block's __shared__ array: thread (0,0,0) runs first and fills it
        for tx in range(B):
            for ty in range(B):
                for tz in range(B):
                    m.run(tree, dict(d_pos=d_pos, d_net_force=d_f, gridX=g[0], gridY=g[1], gridZ=g[2], group_size=len(pos), Nx=Nx, Ny=Ny,
                                     Nz=Nz, d_group_members=members, box=BoxDim(*box), P=P, gridh=gridh, xi=xi, eta=eta, prefac=prefac,
                                     expfac=expfac, shared=shared, blockDim=Vec("dim3", B, B, B), blockIdx=Vec("dim3", p, 0, 0),
                                     threadIdx=Vec("dim3", tx, ty, tz)))
    return np.array([[v.f["x"] for v in g[c]] for c in range(3)]).reshape(3, *grid)


# ---------------------------------------------------------------------------------------------- real-space table + pair formula (K9)
class LazyTable:
    """m_ewaldC1 as PSEv1/Stokes.cc:334-422 fills it -- entry kk holds (Imrr, rr) at r = (kk + 1) dr and at r + dr -- evaluated on
    demand from the closed forms' text (the table has rcut / dr entries; the kernel reads a handful)."""

    def __init__(self, xi, dr):
        import re
        src = "".join(open(os.path.join(REF, "PSEv1/Stokes.cc")).readlines()[343:408])
        found = re.findall(r"\b(Imrr|rr)\s*=\s*(.*?);", src, flags=re.S)
        found = [(n, e) for n, e in found if e.strip() not in ("0", "0, rr = 0")]
        assert [n for n, _ in found] == ["Imrr", "rr"] * 3
        self.expr = {"gt": (found[0][1], found[1][1]), "eq": (found[2][1], found[3][1]), "lt": (found[4][1], found[5][1])}
        self.m = Machine(BUILTINS)
        self.xi, self.dr, self.cache = xi, dr, {}

    def fg(self, r):
        b = "gt" if r > 2.0 else ("eq" if r == 2.0 else "lt")
        env = {"xi": self.xi, "r": r, "Pi": 3.141592653589793, "a": 1.0}
        return self.m.evaluate(self.expr[b][0], env), self.m.evaluate(self.expr[b][1], env)

    def __getitem__(self, kk):
        if kk not in self.cache:
            f0, g0 = self.fg(float(kk) * self.dr + self.dr)
            f1, g1 = self.fg(float(kk + 1) * self.dr + self.dr)
            self.cache[kk] = Vec("Scalar4", f0, g0, f1, g1)
        return self.cache[kk]


def run_mreal(pos, force, nlist, box, xi, rcut, dr, self_mob):
    body = function_source(read("PSEv1/Mobility.cu"), "gpu_stokes_Mreal_kernel")[1]
    m = Machine(BUILTINS)
    tree = m.parse(body)
    n = len(pos)
    d_pos = [Vec("Scalar4", *p, 0.0) for p in pos]
    d_f = [Vec("Scalar4", *f, 0.0) for f in force]
    d_vel = [Vec("Scalar4") for _ in range(n)]
    head, flat = [], []
    for i in range(n):
        head.append(len(flat)); flat += nlist[i]
    table = LazyTable(xi, dr)
    ewald_n = int(rcut / dr - 1)                               # PSEv1/Stokes.cc:310
    for tid in range(n):
        m.run(tree, dict(thread_1d(tid), d_pos=d_pos, d_vel=d_vel, d_net_force=d_f, group_size=n, xi=xi, d_ewaldC1=table, self=self_mob,
                         ewald_cut=rcut, ewald_n=ewald_n, ewald_dr=dr, d_group_members=list(range(n)), box=BoxDim(*box),
                         d_n_neigh=[len(x) for x in nlist], d_nlist=flat, d_headlist=head))
    return np.array([[v.f[c] for c in "xyz"] for v in d_vel])


# ---------------------------------------------------------------------------------------------- Euler step (K15)
def run_step_one(pos, vel, mass, force, image, box, dt, shear_rate):
    body = function_source(read("PSEv1/Stokes.cu"), "gpu_stokes_step_one_kernel")[1]
    m = Machine(BUILTINS)
    tree = m.parse(body)
    n = len(pos)
    d_pos = [Vec("Scalar4", *p, 7.0) for p in pos]
    d_vel = [Vec("Scalar4", *v, mass) for v in vel]
    d_f = [Vec("Scalar4", *f, 0.0) for f in force]
    d_acc = [Vec("Scalar3") for _ in range(n)]
    d_img = [Vec("int3", *im) for im in image]
    for tid in range(n):
        m.run(tree, dict(thread_1d(tid), d_pos=d_pos, d_vel=d_vel, d_accel=d_acc, d_image=d_img, d_group_members=list(range(n)),
                         group_size=n, box=BoxDim(*box), deltaT=dt, d_net_force=d_f, shear_rate=shear_rate))
    return (np.array([[v.f[c] for c in "xyz"] for v in d_pos]), np.array([[v.f[c] for c in "xyz"] for v in d_img]),
            np.array([[v.f[c] for c in "xyz"] for v in d_acc]), [v.f["w"] for v in d_pos])


# ---------------------------------------------------------------------------------------------- the fixture
def engine_params(box, xi, error, max_strain, grid, P=None):
    """eta (and P) as the parameter rule derives them (PSEv1/Stokes.cc:217-236 through oracle/pse_port.py, itself held to the
    reference's expressions by tests/golden/reference_arithmetic.json): an engine created with the same arguments works with
    the same numbers, so the device can be held to these cases too."""
    from oracle import pse_port as pp
    return pp.select_params(box, xi, error, max_strain, grid=grid, P=P)


def main():
    if not os.path.isdir(os.path.join(REF, "PSEv1")):
        sys.exit(f"{REF}/PSEv1 not found: this script only runs where the reference tree is mounted")
    rng = np.random.default_rng(20251003)
    fx = {"generator": "tests/golden/make_kernel_fixture.py + tests/golden/cmini.py"}

    # K1: wave vectors and scale factor, every node of small even / odd / mixed grids, sheared and not
    kcases = []
    for grid, box, xi, eta in [((6, 5, 4), (7.0, 6.0, 5.5, 0.3), 0.5, 0.6), ((4, 4, 4), (5.0, 5.0, 5.0, 0.0), 0.7, 0.5),
                               ((5, 3, 7), (6.0, 4.0, 8.0, -0.45), 0.4, 0.8), ((8, 6, 6), (9.0, 9.5, 8.0, 0.2), 0.55, 0.72)]:
        row = {"grid": grid, "box": box, "xi": xi, "eta": eta}
        for tag, ex in (("as_written", False), ("exact_pi", True)):
            row[tag] = gridk_array(run_setgridk(grid, box, xi, eta, ex), grid).tolist()
        kcases.append(row)
    fx["setgridk"] = {"source": "PSEv1/Helper.cu:285-332 (gpu_stokes_SetGridk_kernel), every node of the full grid: (kx, ky, kz, w)", "cases": kcases}
    # ... and on grids an engine can be created with (eta from the parameter rule): the nodes with a Nyquist index plus a sample
    dcases = []
    for grid, box, xi, error, ms in [((16, 18, 20), (11.0, 12.0, 13.0, 0.3), 0.6, 1e-3, 0.5), ((17, 15, 16), (12.0, 10.0, 11.0, -0.2), 0.5, 1e-4, 0.3)]:
        p = engine_params(box, xi, error, ms, grid)
        g = gridk_array(run_setgridk(grid, box, xi, p["eta"], True), grid)
        i, j, k = np.meshgrid(*[np.arange(n) for n in grid], indexing="ij")
        pick = ((grid[0] % 2 == 0) & (i == grid[0] // 2)) | ((grid[1] % 2 == 0) & (j == grid[1] // 2)) | ((grid[2] % 2 == 0) & (k == grid[2] // 2))
        pick |= (i + j + k == 0) | (rng.uniform(size=i.shape) < 0.03)
        nodes = np.stack([i[pick], j[pick], k[pick]], axis=1)
        dcases.append({"grid": grid, "box": box, "xi": xi, "error": error, "max_strain": ms, "eta": p["eta"], "P": p["P"],
                       "nodes": nodes.tolist(), "k": g[pick].tolist()})
    fx["setgridk_engine"] = {"source": "PSEv1/Helper.cu:285-332 with exact pi; nodes (i, j, k) and their (kx, ky, kz, w)", "cases": dcases}

    # K5 + K6 on top of the exact-pi wave vectors
    gcases, bcases = [], []
    for grid, box, xi, eta in [((6, 5, 4), (7.0, 6.0, 5.5, 0.3), 0.5, 0.6), ((4, 4, 4), (5.0, 5.0, 5.0, 0.0), 0.7, 0.5),
                               ((5, 3, 7), (6.0, 4.0, 8.0, -0.45), 0.4, 0.8), ((6, 8, 5), (7.5, 8.0, 6.0, 0.35), 0.6, 0.7)]:
        gridk = run_setgridk(grid, box, xi, eta, True)
        real = rng.normal(size=(3,) + grid)
        fhat = np.fft.fftn(real, axes=(1, 2, 3))                                # the spectrum of a REAL field, as the spread delivers it
        out = run_green(gridk, fhat, grid)
        back = np.fft.ifftn(out, axes=(1, 2, 3)) * np.prod(grid)                # unnormalised inverse; the reference reads the real part (Mobility.cu:447)
        gcases.append({"grid": grid, "box": box, "xi": xi, "eta": eta, "real_in": real.tolist(), "real_out": back.real.tolist(),
                       "max_imag_out": float(np.abs(back.imag).max())})
        kT, dt, seed, ts = 1.3, 2e-3, 11, 5
        h3 = box[0] / grid[0] * box[1] / grid[1] * box[2] / grid[2]
        noise = run_brownian_grid(gridk, grid, kT, dt, h3, seed, ts)
        field = np.fft.ifftn(noise, axes=(1, 2, 3)) * np.prod(grid)
        bcases.append({"grid": grid, "box": box, "xi": xi, "eta": eta, "kT": kT, "dt": dt, "seed": seed, "timestep": ts,
                       "real_out": field.real.tolist(), "max_imag_out": float(np.abs(field.imag).max()),
                       "nodes_written": int(np.count_nonzero(np.abs(noise).sum(axis=0)))})
    fx["green"] = {"source": "PSEv1/Mobility.cu:264-299 applied to the FFT of real_in with the exact-pi wave vectors; real_out = real part of the "
                             "unnormalised inverse FFT", "cases": gcases}
    fx["brownian_grid"] = {"source": "PSEv1/Brownian.cu:153-345 on zeroed grids, every thread; the Saru stand-in returns the port's Philox "
                                     "uniforms of the thread's node; real_out = real part of the unnormalised inverse FFT", "cases": bcases}

    # K3: supports, wrap, sheared node positions, weights -- parameters an engine derives too; only the nodes that were written
    scases = []
    for P, grid, box, xi, error, ms in [(4, (16, 18, 20), (11.0, 12.0, 13.0, 0.0), 0.6, 1e-3, 0.0), (5, (18, 16, 20), (12.0, 11.0, 13.0, 0.25), 0.6, 1e-3, 0.3),
                                        (6, (16, 16, 18), (10.0, 10.5, 11.0, -0.4), 0.5, 1e-3, 0.5), (7, (18, 20, 16), (11.0, 12.0, 10.0, 0.5), 0.5, 1e-4, 0.5),
                                        (13, (28, 26, 30), (14.0, 13.0, 15.0, 0.2), 0.5, 1e-6, 0.5), (6, (12, 10, 14), (9.0, 8.0, 10.0, 0.3), 0.6, 1e-3, 0.5)]:
        p = engine_params(box, xi, error, ms, grid, P)
        n = {13: 3, 7: 4}.get(P, 5)                                             # P^3 nodes per particle are stored
        frac = rng.uniform(size=(n, 3))
        frac[0] = [0.001, 0.999, 0.5]                                           # supports that wrap around both ends
        frac[1] = (np.floor(frac[1] * grid) + [0.4999999, 0.5000001, 0.0]) / grid  # either side of half-way between nodes (odd-P centring rule;
        frac[2] = (np.floor(frac[2] * grid) + [0.5000001, 0.25, 0.4999999]) / grid  # an exact tie is decided by rounding, in any arithmetic)
        y = (frac[:, 1] - 0.5) * box[1]
        pos = np.stack([(frac[:, 0] - 0.5) * box[0] + box[3] * y, y, (frac[:, 2] - 0.5) * box[2]], axis=1)
        force = rng.normal(size=(n, 3))
        g = run_spread(pos, force, grid, box, P, xi, p["eta"]).reshape(3, -1)
        nz = np.nonzero(np.abs(g).sum(axis=0))[0]
        scases.append({"P": P, "grid": grid, "box": box, "xi": xi, "error": error, "max_strain": ms, "eta": p["eta"], "pos": pos.tolist(),
                       "force": force.tolist(), "nodes": nz.tolist(), "values": g[:, nz].T.tolist()})
    fx["spread"] = {"source": "PSEv1/Mobility.cu:114-252, one block per particle, all threads; nodes = linear indices (x Ny + y) Nz + z of the nodes "
                              "written, values = real parts of the three grids there (all other nodes are zero)", "cases": scases}

    # K9: pair formula; table spacing as written (1e-3) and 1e-6 (the interpolation error then vanishes)
    box = (30.0, 28.0, 32.0, 0.3)
    xi, rcut = 0.5, 5.2565
    pos = np.array([[0.0, 0.0, 0.0], [1.2, 0.3, 0.0], [-2.0, 0.0, 0.0], [0.4, 3.1, -1.5], [0.05, 0.02, 0.01], [0.0, 0.0, 5.2],
                    [2.5 + 0.3 * 28.0, -1.0 + 28.0, 0.7], [-14.9, 1.0, 15.5]])    # overlapping, touching (r = 2), separated, nearly coincident, images
    force = rng.normal(size=(len(pos), 3))
    nlist = [[j for j in range(len(pos)) if j != i] for i in range(len(pos))]   # the kernel itself applies the cutoff
    selfm = (1.0 + 4.0 * math.sqrt(math.pi) * xi * math.erfc(2.0 * xi) - math.exp(-4.0 * xi * xi)) / (4.0 * math.sqrt(math.pi) * xi)
    fx["mreal"] = {"source": "PSEv1/Mobility.cu:594-687 with the table of PSEv1/Stokes.cc:334-422 (closed forms :348-406)", "box": box, "xi": xi,
                   "rcut": rcut, "self": selfm, "pos": pos.tolist(), "force": force.tolist(),
                   "vel_dr_1e-3": run_mreal(pos, force, nlist, box, xi, rcut, 1e-3, selfm).tolist(),
                   "vel_dr_1e-6": run_mreal(pos, force, nlist, box, xi, rcut, 1e-6, selfm).tolist()}

    # K15
    box = (20.0, 18.0, 22.0, 0.35)
    n = 12
    pos = (rng.uniform(size=(n, 3)) - 0.5) * [20.0, 18.0, 22.0] * 0.999
    pos[:, 0] += 0.35 * pos[:, 1]
    vel = rng.normal(size=(n, 3)) * 40.0                                        # large enough to cross every face within dt
    force = rng.normal(size=(n, 3))
    image = rng.integers(-2, 3, size=(n, 3))
    dt, rate, mass = 0.05, 0.7, 2.5
    p2, im2, acc, w = run_step_one(pos, vel, mass, force, image, box, dt, rate)
    fx["step_one"] = {"source": "PSEv1/Stokes.cu:137-192", "box": box, "dt": dt, "shear_rate": rate, "mass": mass, "pos": pos.tolist(),
                      "vel": vel.tolist(), "force": force.tolist(), "image": image.tolist(), "pos_out": p2.tolist(), "image_out": im2.tolist(),
                      "accel_out": acc.tolist(), "pos_w_kept": w}
    import gzip
    with gzip.GzipFile(OUT, "wb", mtime=0) as f:                                 # mtime 0: the same numbers give the same bytes
        f.write(json.dumps(fx).encode())
    print("written", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
