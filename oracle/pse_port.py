"""
pse_port.py -- TEST INFRASTRUCTURE ONLY.  NumPy restatement ("port") of the reference's PSE
algorithm, step for step, in fp64.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; the product path (pse_amd/) never does.

Where pse_oracle.c evaluates the *exact* periodic RPY mobility (direct Ewald sums), this file
restates the *approximate* fast algorithm the reference actually runs on the GPU, so the HIP path
can be compared against it at round-off tolerance (same grid, same P, same eta, same RNG):

  select_params      PSEv1/Stokes.cc:129-236   (rcut, kmax, grid 2^a3^b5^c, lambda, gaussm, P, eta)
  spread             PSEv1/Mobility.cu:114-252 (+ prefac/expfac PSEv1/Brownian.cu:826-829)
  wave_scale         PSEv1/Helper.cu:285-332 (sheared k, exact pi) x PSEv1/Mobility.cu:264-299
  noise_k            PSEv1/Brownian.cu:153-345 (variances, Hermitian symmetry, Nyquist x sqrt2)
  gather             PSEv1/Mobility.cu:325-477 (+ weight h^3*prefac, PSEv1/Brownian.cu:872)
  mobility_real      PSEv1/Mobility.cu:594-687 (closed-form f,g instead of the fp32 table)
  psi_particles      PSEv1/Brownian.cu:99-130  (uniform on (-sqrt3, sqrt3))
  lanczos_sqrt       PSEv1/Brownian.cu:357-765 (Chow & Saad; step-norm stopping rule)
  brownian_velocity  PSEv1/Brownian.cu:772-923 (the combined deterministic + stochastic step)
  integrate          PSEv1/Stokes.cu:137-192   (Euler + shear + triclinic wrap)
  shear functions    PSEv1/SpecificShearFunction.h:16-223, PSEv1/VariantShearFunction.{h:46-48,cc:34-43}

Deliberate differences from the reference: fp64 throughout; exact pi;
real-to-complex half spectrum; random numbers from Philox4x32-10 keyed (seed, timestep, index,
domain) instead of HOOMD's Saru (not in the reference tree -> RNG stream parity is unpinned;
the *distribution* -- uniform, variance-matched -- is the reference's).

Parity pin (all fixtures are numbers generated in the build container from the reference's own text; no text travels):
  * select_params, hash_seed, the k-space factor, the real-space closed forms and the shear classes: the values the reference's
    expression text takes (tests/golden/reference_arithmetic.json <- make_reference_fixture.py; tests/test_reference_pin.py);
  * spread, kvectors / wave_scale, noise_k, the pair formula, integrate: the bodies of the reference's kernels executed thread by
    thread by the C-subset interpreter tests/golden/cmini.py (reference_kernels.json.gz <- make_kernel_fixture.py;
    tests/test_reference_kernels.py);
  * gather, and the constants and order of the combined step: gpu_stokes_Contract_kernel run with real barriers, the host wrapper
    run with recording stubs (reference_driver.json.gz <- make_driver_fixture.py; tests/test_reference_driver.py);
  * lanczos_sqrt: the host driver gpu_stokes_BrealLanczos_wrap executed on dense operators -- same m, same vector
    (reference_lanczos.json <- make_lanczos_fixture.py; tests/test_reference_lanczos.py).
Unpinned by construction: the random STREAM (Saru is not in the reference tree).
"""
import ctypes
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    """ctypes handle of the C oracle (built by `make -C oracle`)."""
    global _LIB
    if _LIB is None:
        # PSE_ASAN_DIR: the -fsanitize=address,undefined build of this same source (tools/asan.py builds it)
        d = os.environ.get("PSE_ASAN_DIR")
        if d and not os.path.exists(os.path.join(d, ".pse_asan_build")):   # a stray variable: not a sanitizer build (pse_amd/_lib.py asan_dir)
            d = None
        path = os.path.join(d or _HERE, "libpse_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(path)
        L.pse_oracle_self.restype = ctypes.c_double
        L.pse_oracle_self.argtypes = [ctypes.c_double]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


# ------------------------------------------------------------------------------------------ C oracle wrappers
def fg_real(r, xi):
    r = np.atleast_1d(np.asarray(r, float))
    f = np.empty_like(r); g = np.empty_like(r)
    fo = ctypes.c_double(); go = ctypes.c_double()
    L = lib()
    for i, ri in enumerate(r):
        L.pse_oracle_fg_real(ctypes.c_double(ri), ctypes.c_double(xi), ctypes.byref(fo), ctypes.byref(go))
        f[i] = fo.value; g[i] = go.value
    return f, g


def fg_wave(r, xi, quad=False):
    fo = ctypes.c_double(); go = ctypes.c_double()
    fn = lib().pse_oracle_fg_wave_quad if quad else lib().pse_oracle_fg_wave
    fn(ctypes.c_double(r), ctypes.c_double(xi), ctypes.byref(fo), ctypes.byref(go))
    return fo.value, go.value


def self_mobility(xi):
    return lib().pse_oracle_self(float(xi))


def mobility_direct(pos, force, box, xi, tol=1e-14, parts=3, nthreads=0):
    """Exact periodic RPY U = M.F by direct Ewald summation (O(N^2))."""
    pos = np.ascontiguousarray(pos, float); force = np.ascontiguousarray(force, float)
    box = np.ascontiguousarray(box, float); out = np.zeros_like(pos)
    lib().pse_oracle_mobility_direct(len(pos), _p(pos), _p(force), _p(box), ctypes.c_double(xi),
                                     ctypes.c_double(tol), int(parts), _p(out), int(nthreads))
    return out


def mobility_dense(pos, box, xi, tol=1e-14, parts=3, nthreads=0):
    pos = np.ascontiguousarray(pos, float); box = np.ascontiguousarray(box, float)
    n = len(pos); M = np.zeros((3 * n, 3 * n))
    lib().pse_oracle_mobility_dense(n, _p(pos), _p(box), ctypes.c_double(xi), ctypes.c_double(tol),
                                    int(parts), _p(M), int(nthreads))
    return M


def mobility_real(pos, force, box, xi, rcut, nthreads=0, rounded=False):
    """Near-field sum as the reference's Mreal kernel does it: minimum image, r < rcut, + self.
    rounded: with the pair coefficients as the build's Lanczos mat-vecs read them from their 16-byte records (oracle/pse_oracle.c pair_term; the
    deterministic M.F is always the double-precision sum)."""
    pos = np.ascontiguousarray(pos, float); force = np.ascontiguousarray(force, float)
    box = np.ascontiguousarray(box, float); out = np.zeros_like(pos)
    fn = lib().pse_oracle_mreal_cutoff_rounded if rounded else lib().pse_oracle_mreal_cutoff
    fn(len(pos), _p(pos), _p(force), _p(box), ctypes.c_double(xi), ctypes.c_double(rcut), _p(out), int(nthreads))
    return out


def mobility_real_rows(pos, force, box, xi, rcut, rows, nthreads=0):
    """Rows `rows` of the near-field sum (for parity checks at sizes where all rows would take too long)."""
    pos = np.ascontiguousarray(pos, float); force = np.ascontiguousarray(force, float)
    box = np.ascontiguousarray(box, float); rows = np.ascontiguousarray(rows, np.int32)
    out = np.zeros((len(rows), 3))
    lib().pse_oracle_mreal_cutoff_rows(len(pos), _p(pos), _p(force), _p(box), ctypes.c_double(xi), ctypes.c_double(rcut),
                                       len(rows), rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _p(out), int(nthreads))
    return out


def max_threads():
    return lib().pse_oracle_max_threads()


# ------------------------------------------------------------------------------------------ parameters
def _next235(n):
    best = None
    p2 = 1
    while p2 <= 4096:
        p3 = p2
        while p3 <= 4096:
            p5 = p3
            while p5 <= 4096:
                if p5 >= max(n, 8) and (best is None or p5 < best):
                    best = p5
                p5 *= 5
            p3 *= 3
        p2 *= 2
    if best is None:
        raise ValueError("grid dimension beyond 4096")
    return best


def select_params(box, xi=0.5, error=1e-3, max_strain=0.5, grid=None, P=None, rcut=None):
    """PSEv1/Stokes.cc:129-236 in fp64.  box = (Lx, Ly, Lz, xy).  grid/P/rcut are explicit overrides."""
    Lx, Ly, Lz = float(box[0]), float(box[1]), float(box[2])
    s = math.sqrt(-math.log(error))
    p = {"xi": float(xi), "error": float(error), "max_strain": float(max_strain)}
    p["rcut"] = s / xi if rcut is None else float(rcut)                       # Stokes.cc:135
    kmax = int(2.0 * s * xi) + 1                                              # Stokes.cc:138
    if grid is None:
        grid = tuple(_next235(int(kmax * L / math.pi) + 1) for L in (Lx, Ly, Lz))   # Stokes.cc:143-199
    p["grid"] = tuple(int(g) for g in grid)
    g2 = max_strain * max_strain
    lam = 1.0 + g2 / 2.0 + max_strain * math.sqrt(1.0 + g2 / 4.0)            # Stokes.cc:217-219
    i = 0
    while math.erfc((1.0 + 0.01 * i) / math.sqrt(2.0 * lam)) > error:         # Stokes.cc:225-228
        i += 1
    gaussm = 1.0 + 0.01 * i
    if P is None:
        P = int(gaussm * gaussm / math.pi) + 1                                # Stokes.cc:229
        P = min(P, *p["grid"])                                                # Stokes.cc:231-233
    p["lambda"] = lam; p["gaussm"] = gaussm; p["P"] = int(P)
    h = (Lx / p["grid"][0], Ly / p["grid"][1], Lz / p["grid"][2])
    w = p["P"] * h[0] / 2.0
    p["eta"] = (2.0 * w / gaussm) ** 2 * xi * xi                              # Stokes.cc:234-236
    p["h"] = h
    p["self"] = self_mobility(xi)
    p["ewald_n"] = int(p["rcut"] / 0.001 - 1)                                  # Stokes.cc:309-310 (reference table size)
    return p


# ------------------------------------------------------------------------------------------ geometry helpers
def fractional(pos, box):
    """f in [0,1)^3 with the box centred on the origin (HOOMD BoxDim::makeFraction as used at Mobility.cu:173)."""
    Lx, Ly, Lz, xy = box
    f = np.empty_like(pos)
    f[:, 0] = (pos[:, 0] - xy * pos[:, 1]) / Lx + 0.5
    f[:, 1] = pos[:, 1] / Ly + 0.5
    f[:, 2] = pos[:, 2] / Lz + 0.5
    return f - np.floor(f)


def _support(pos, box, p):
    """Per particle: node indices (N,P) per axis (wrapped) and lattice offsets Delta (N,P) in grid units."""
    grid, P = p["grid"], p["P"]
    f = fractional(pos, box)
    idx = []; dlt = []
    for a in range(3):
        s = f[:, a] * grid[a]
        i0 = np.floor(s).astype(np.int64)
        start = i0 - P // 2 + 1 - (P % 2) * ((s - i0) < 0.5)                  # Mobility.cu:212-214
        t = start[:, None] + np.arange(P)[None, :]
        dlt.append(t - s[:, None])
        idx.append(np.mod(t, grid[a]))                                        # Mobility.cu:217-219
    return idx, dlt


def _weights(pos, box, p):
    Lx, Ly, Lz, xy = box
    hx, hy, hz = p["h"]
    idx, dlt = _support(pos, box, p)
    dx = hx * dlt[0][:, :, None] + xy * hy * dlt[1][:, None, :]              # (N,P,P): Mobility.cu:223-230
    dy = hy * dlt[1]
    dz = hz * dlt[2]
    expfac = 2.0 * p["xi"] ** 2 / p["eta"]                                    # Brownian.cu:829
    prefac = (2.0 * p["xi"] ** 2 / math.pi / p["eta"]) ** 1.5                 # Brownian.cu:828
    r2 = dx[:, :, :, None] ** 2 + (dy ** 2)[:, None, :, None] + (dz ** 2)[:, None, None, :]
    w = prefac * np.exp(-expfac * r2)                                         # (N,P,P,P)
    lin = (idx[0][:, :, None, None] * p["grid"][1] + idx[1][:, None, :, None]) * p["grid"][2] + idx[2][:, None, None, :]
    return w, lin


def spread(pos, force, box, p):
    w, lin = _weights(pos, box, p)
    ng = int(np.prod(p["grid"]))
    out = np.zeros((3, ng))
    for c in range(3):
        np.add.at(out[c], lin.ravel(), (w * force[:, c][:, None, None, None]).ravel())
    return out.reshape((3,) + p["grid"])


def gather(ugrid, pos, box, p):
    w, lin = _weights(pos, box, p)
    h3 = p["h"][0] * p["h"][1] * p["h"][2]
    out = np.empty((len(pos), 3))
    for c in range(3):
        out[:, c] = h3 * np.sum(w * ugrid[c].ravel()[lin], axis=(1, 2, 3))   # Brownian.cu:872
    return out


def kvectors(box, p, full=False):
    """Sheared wave vectors on the half spectrum (Helper.cu:300-315, exact pi) and scale w(k) (Helper.cu:318-327).
    full: every node of the C2C grid the reference works on (kz index 0..Nz-1) instead of kz <= Nz/2."""
    Lx, Ly, Lz, xy = box
    Nx, Ny, Nz = p["grid"]
    i = np.arange(Nx); i = np.where(i < (Nx + 1) // 2, i, i - Nx).astype(float)
    j = np.arange(Ny); j = np.where(j < (Ny + 1) // 2, j, j - Ny).astype(float)
    k = np.arange(Nz if full else Nz // 2 + 1)
    k = np.where(k < (Nz + 1) // 2, k, k - Nz).astype(float)   # the z-Nyquist plane folds to -Nz/2 like x and y (Helper.cu:312)
    kx = 2 * math.pi * i[:, None, None] / Lx + 0 * j[None, :, None] + 0 * k[None, None, :]
    ky = 2 * math.pi * (j[None, :, None] - xy * i[:, None, None] * Ly / Lx) / Ly + 0 * k[None, None, :]
    kz = 2 * math.pi * k[None, None, :] / Lz + 0 * kx
    k2 = kx * kx + ky * ky + kz * kz
    q = k2 / (4.0 * p["xi"] ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = 6.0 * math.pi * (1.0 + q) * np.exp(-(1.0 - p["eta"]) * q) / k2 / float(Nx * Ny * Nz)
        sinc = np.sin(np.sqrt(k2)) / np.sqrt(k2)
    w[0, 0, 0] = 0.0; sinc[0, 0, 0] = 0.0
    return kx, ky, kz, k2, w, sinc


def _project(kx, ky, kz, k2, v):
    with np.errstate(divide="ignore", invalid="ignore"):
        kd = (kx * v[0] + ky * v[1] + kz * v[2]) / k2
    kd[0, 0, 0] = 0.0
    return np.stack([v[0] - kx * kd, v[1] - ky * kd, v[2] - kz * kd])


def _apply_symmetrised(box, p, v, scale, y0=0, nyl=None):
    """v(k) -> 1/2 [ s(k) (I - kk) + s(k') (I - k'k') ] v(k) on the half spectrum, k' the wave vector of the node's conjugate
    partner (-i, -j, -k) mod N as the index folding gives it; scale(w, sinc) -> s.

    The reference fills the FULL complex grid node by node -- every node with the folded wave vector of its own index
    (Helper.cu:307-315) -- transforms back with a C2C FFT and reads the real part (Mobility.cu:447).  Taking the real part
    keeps the Hermitian part of the spectrum, 1/2 (G(k) + conj G(partner)).  Where an index is a Nyquist index (even N) the
    folding gives node and partner the SAME sign in that component, k' != -k, and the two operators differ; everywhere else
    k' = -k, the operator is even in k, and nothing changes.  Pinned by tests/golden/reference_kernels.json.gz.
    y0, nyl: v holds the y rows [y0, y0 + nyl) only (a slab rank after the transpose)."""
    Nx, Ny, Nz = p["grid"]
    Nzh = Nz // 2 + 1
    nyl = Ny if nyl is None else nyl
    kx, ky, kz, k2, w, sinc = kvectors(box, p, full=True)
    rows = np.arange(y0, y0 + nyl)

    def part(sel):
        with np.errstate(divide="ignore", invalid="ignore"):
            kd = (kx[sel] * v[0] + ky[sel] * v[1] + kz[sel] * v[2]) / k2[sel]
        if y0 == 0:
            kd[0, 0, 0] = 0.0
        return np.stack([v[0] - kx[sel] * kd, v[1] - ky[sel] * kd, v[2] - kz[sel] * kd]) * scale(w[sel], sinc[sel])
    own = part(np.ix_(np.arange(Nx), rows, np.arange(Nzh)))
    par = part(np.ix_((-np.arange(Nx)) % Nx, (-rows) % Ny, (-np.arange(Nzh)) % Nz))
    return 0.5 * (own + par)


def wave_scale(fhat, box, p, y0=0, nyl=None):
    return _apply_symmetrised(box, p, fhat, lambda w, sinc: w * sinc * sinc, y0, nyl)   # Mobility.cu:283-295


# ------------------------------------------------------------------------------------------ RNG (Philox4x32-10)
_M0, _M1, _W0, _W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
KEY1 = 0x50534531  # 'PSE1'
DOMAIN_PARTICLE, DOMAIN_GRID_A, DOMAIN_GRID_B = 0, 1, 2


def philox4x32(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10; counters are uint32 arrays (broadcast), returns 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(*[np.asarray(c, np.uint32) for c in (c0, c1, c2, c3)])
    c0, c1, c2, c3 = c0.copy(), c1.copy(), c2.copy(), c3.copy()
    k0 = np.uint32(k0); k1 = np.uint32(k1)
    m32 = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64); p1 = _M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32); lo0 = (p0 & m32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32); lo1 = (p1 & m32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF); k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _u(x, s):
    """uint32 -> uniform on (-s, s)."""
    return ((x.astype(np.float64) + 0.5) * (2.0 ** -32) * 2.0 - 1.0) * s


def psi_particles(n, seed, timestep):
    """Brownian.cu:99-130: three uniforms on (-sqrt3, sqrt3) per particle (variance 1), keyed by particle index."""
    idx = np.arange(n, dtype=np.uint64)
    r = philox4x32(idx & np.uint64(0xFFFFFFFF), idx >> np.uint64(32), timestep, DOMAIN_PARTICLE, seed, KEY1)
    s = math.sqrt(3.0)
    return np.stack([_u(r[0], s), _u(r[1], s), _u(r[2], s)], axis=1)


def noise_k(box, p, kT, dt, seed, timestep):
    """Brownian.cu:153-345 on the half spectrum: sqrt(2kT/(dt h^3)) sqrt(w) sinc (I - kk) psi_k."""
    Nx, Ny, Nz = p["grid"]; Nzh = Nz // 2 + 1
    i = np.arange(Nx)[:, None, None]; j = np.arange(Ny)[None, :, None]; k = np.arange(Nzh)[None, None, :]
    i, j, k = np.broadcast_arrays(i, j, k)
    own = (i * Ny + j) * Nz + k
    on_plane = (k == 0) | ((Nz % 2 == 0) & (k == Nz // 2))
    ic = (Nx - i) % Nx; jc = (Ny - j) % Ny
    partner = (ic * Ny + jc) * Nz + k
    selfc = on_plane & (partner == own)
    canon = np.where(on_plane, np.minimum(own, partner), own).astype(np.uint64)
    flip = on_plane & (partner < own)
    lo = canon & np.uint64(0xFFFFFFFF); hi = canon >> np.uint64(32)
    a = philox4x32(lo, hi, timestep, DOMAIN_GRID_A, seed, KEY1)
    b = philox4x32(lo, hi, timestep, DOMAIN_GRID_B, seed, KEY1)
    s = math.sqrt(1.5)                                                        # Brownian.cu:178-189: var 1/2 each
    re = [_u(a[0], s), _u(a[1], s), _u(a[2], s)]
    im = [_u(a[3], s), _u(b[0], s), _u(b[1], s)]
    psi = []
    for c in range(3):
        z = re[c] + 1j * np.where(flip, -im[c], im[c])
        z = np.where(selfc, math.sqrt(2.0) * re[c] + 0j, z)                   # Brownian.cu:255-268
        psi.append(z)
    h3 = p["h"][0] * p["h"][1] * p["h"][2]
    fac = math.sqrt(2.0 * kT / dt / h3)                                       # Brownian.cu:197
    return _apply_symmetrised(box, p, np.stack(psi), lambda w, sinc: fac * np.sqrt(w) * sinc)


# ------------------------------------------------------------------------------------------ mobility
def mobility_wave(pos, force, box, p, extra_k=None):
    fh = np.fft.rfftn(spread(pos, force, box, p), axes=(1, 2, 3))
    uh = wave_scale(fh, box, p)
    if extra_k is not None:
        uh = uh + extra_k
    ug = np.fft.irfftn(uh, s=p["grid"], axes=(1, 2, 3), norm="forward")     # unnormalised inverse, 1/Ng lives in w
    return gather(ug, pos, box, p)


def mobility(pos, force, box, p):
    return mobility_wave(pos, force, box, p) + mobility_real(pos, force, box, p["xi"], p["rcut"])


def lanczos_sqrt(matvec, psi, m_in=2, tol=1e-3, m_max=100):
    """M^{1/2} psi by Lanczos (Brownian.cu:440-739).  Returns (vector, m).  Stops at the first m >= m_in whose
    step norm ||u_m - u_{m-1}|| / sqrt(psi.M.psi/|psi|^2) <= tol (Brownian.cu:604-724); since the basis is
    orthonormal the step norm is evaluated on the small vectors t_m."""
    shape = psi.shape
    psi = psi.ravel()
    norm = np.linalg.norm(psi)
    V = [psi / norm]; alpha = []; beta = [0.0]
    vjm1 = np.zeros_like(psi)

    def tvec(m):
        T = np.diag(alpha[:m]) + np.diag(beta[1:m], 1) + np.diag(beta[1:m], -1)
        lam, W = np.linalg.eigh(T)
        return W @ (np.sqrt(np.maximum(lam, 0.0)) * W[0, :])

    t_prev = None
    m = 0
    while True:
        vj = V[m]
        v = matvec(vj.reshape(shape)).ravel() - beta[m] * vjm1
        a = vj @ v; v = v - a * vj; b = np.linalg.norm(v)
        alpha.append(a); beta.append(b); m += 1
        if b < 1e-8:                      # invariant subspace: exact (reference would drop this vector, Brownian.cu:503-506)
            t = tvec(m); break
        vjm1 = vj; V.append(v / b)
        start = max(m_in - 1, 1)
        if m >= start:
            t = tvec(m)
            if t_prev is not None:
                step = np.linalg.norm(t - np.append(t_prev, 0.0)) / math.sqrt(alpha[0])
                if step <= tol or m >= m_max:
                    break
            t_prev = t
    u = np.zeros_like(psi)
    for q in range(m):
        u += t[q] * V[q]
    return (norm * u).reshape(shape), m


def brownian_velocity(pos, force, box, p, kT, dt, seed, timestep, m_in=2, pair_rounded=True):
    """Brownian.cu:772-923: u = M.F + sqrt(2kT/dt) M^{1/2} psi, wave noise drawn in k-space.
    pair_rounded: the near-field operator inside the Lanczos iteration with the rounded pair coefficients of the build's pair list
    (False: the all-double algorithm the golden fixture was pinned with; the two differ by ~1e-8 relative)."""
    nk = noise_k(box, p, kT, dt, seed, timestep) if kT > 0 else None
    u = mobility_wave(pos, force, box, p, extra_k=nk) + mobility_real(pos, force, box, p["xi"], p["rcut"])
    m = m_in
    if kT > 0:
        psi = psi_particles(len(pos), seed, timestep)
        zero = np.zeros_like(pos)
        # the operator of the Lanczos iteration carries the rounded pair coefficients of the build's per-step pair list
        mv = lambda v: mobility_real(pos, np.ascontiguousarray(v), box, p["xi"], p["rcut"], rounded=pair_rounded)
        ub, m = lanczos_sqrt(mv, psi, m_in=m_in, tol=p["error"])
        u = u + math.sqrt(2.0 * kT / dt) * ub
    return u, m


def wrap(pos, image, box):
    """Triclinic wrap into the box centred at the origin; y images shift x by xy*Ly (HOOMD convention)."""
    Lx, Ly, Lz, xy = box
    pos = pos.copy(); image = image.copy()
    n = np.floor(pos[:, 2] / Lz + 0.5); pos[:, 2] -= n * Lz; image[:, 2] += n.astype(image.dtype)
    n = np.floor(pos[:, 1] / Ly + 0.5); pos[:, 1] -= n * Ly; pos[:, 0] -= n * xy * Ly; image[:, 1] += n.astype(image.dtype)
    n = np.floor((pos[:, 0] - xy * pos[:, 1]) / Lx + 0.5); pos[:, 0] -= n * Lx; image[:, 0] += n.astype(image.dtype)
    return pos, image


def min_image(d, box):
    """Minimum image of separation vectors d[..., 3] in the xy-tilted cell: wrap z, then y (a y image shifts x by xy*Ly),
    then x -- HOOMD's BoxDim::minImage as used at PSEv1/Mobility.cu:648 (restated: the source is not in the tree).
    Exact for |d_true| < half the smallest cell height and |xy| <= 0.5."""
    Lx, Ly, Lz, xy = box
    d = np.array(d, dtype=float, copy=True)
    n = np.rint(d[..., 2] / Lz); d[..., 2] -= n * Lz
    n = np.rint(d[..., 1] / Ly); d[..., 1] -= n * Ly; d[..., 0] -= n * xy * Ly
    n = np.rint(d[..., 0] / Lx); d[..., 0] -= n * Lx
    return d


def pair_repulsion(pos, box, k, sigma=2.0):
    """Soft repulsion k (sigma - r) r_hat over minimum-image pairs with r < sigma (the force provider of SURVEY.md 8 f4;
    O(N^2), triclinic minimum image as in HOOMD: y images shift x by xy*Ly)."""
    Lx, Ly, Lz, xy = box
    d = min_image(pos[:, None, :] - pos[None, :, :], box)
    r = np.linalg.norm(d, axis=2)
    with np.errstate(divide="ignore", invalid="ignore"):
        c = np.where((r < sigma) & (r > 0.0), k * (sigma - r) / r, 0.0)
    return (c[..., None] * d).sum(axis=1)


def integrate(pos, image, vel, box, dt, shear_rate):
    """Stokes.cu:156-190."""
    v = vel.copy(); v[:, 0] += shear_rate * pos[:, 1]
    return wrap(pos + v * dt, image, box)


# ------------------------------------------------------------------------------------------ shear functions
class SteadyShear:                                                            # SpecificShearFunction.h:47-74
    def __init__(self, shear_rate, offset, dt): self.rate, self.offset, self.dt = shear_rate, offset, dt
    def shear_rate(self, t): return self.rate
    def strain(self, t): return self.rate * (t - self.offset) * self.dt


class SinShear:                                                               # SpecificShearFunction.h:16-46
    def __init__(self, max_rate, freq, offset, dt): self.a, self.f, self.offset, self.dt = max_rate, freq, offset, dt
    def shear_rate(self, t): return self.a * math.cos(self.f * 2 * math.pi * ((t - self.offset) * self.dt))
    def strain(self, t): return self.a * math.sin(self.f * 2 * math.pi * ((t - self.offset) * self.dt)) / self.f / 2 / math.pi


class ChirpShear:                                                             # SpecificShearFunction.h:76-127
    def __init__(self, amp, w0, wf, T, offset, dt): self.amp, self.w0, self.wf, self.T, self.offset, self.dt = amp, w0, wf, T, offset, dt
    def _omega(self, t): return self.w0 * math.exp(self.dt * (t - self.offset) * math.log(self.wf / self.w0) / self.T)
    def _phase(self, t):
        lg = math.log(self.wf / self.w0)
        return self.T * self.w0 / lg * (math.exp(self.dt * (t - self.offset) * lg / self.T) - 1)
    def shear_rate(self, t): return self.amp * self._omega(t) * math.cos(self._phase(t))
    def strain(self, t): return self.amp * math.sin(self._phase(t))


class TukeyWindow:                                                            # SpecificShearFunction.h:129-192
    def __init__(self, T, param, offset, dt):
        self.T, self.p, self.offset, self.dt = T, param, offset, dt
        self.om = 2 * math.pi / param
    def _rel(self, t): return (t - self.offset) * self.dt / self.T
    def shear_rate(self, t):
        r = self._rel(t)
        if r <= 0 or r >= 1 or (self.p / 2 <= r <= 1 - self.p / 2): return 0.0
        if r < 0.5: return -(math.sin(self.om * (r - self.p / 2))) / 2 * self.om / self.T
        return -(math.sin(self.om * (r - 1 + self.p / 2))) / 2 * self.om / self.T
    def strain(self, t):
        r = self._rel(t)
        if r <= 0 or r >= 1: return 0.0
        if self.p / 2 <= r <= 1 - self.p / 2: return 1.0
        if r < 0.5: return (1 + math.cos(self.om * (r - self.p / 2))) / 2
        return (1 + math.cos(self.om * (r - 1 + self.p / 2))) / 2


class Windowed:                                                               # SpecificShearFunction.h:194-223
    def __init__(self, base, window): self.b, self.w, self.offset = base, window, base.offset
    def shear_rate(self, t): return self.b.shear_rate(t) * self.w.strain(t) + self.b.strain(t) * self.w.shear_rate(t)
    def strain(self, t): return self.b.strain(t) * self.w.strain(t)


def variant_value(func, t, total, lo, hi):
    """VariantShearFunction.cc:34-43 / .h:46-48: wrapped strain for the box tilt."""
    rng = hi - lo
    wrapv = lambda x: x - rng * math.floor((x - lo) / rng)
    if t < func.offset: return 0.0
    if t >= func.offset + total: return wrapv(func.strain(func.offset + total))
    return wrapv(func.strain(t))


def hash_seed(seed):
    """PSEv1/Stokes.cc:102 (uint32 arithmetic)."""
    s = (seed * 0x12345677 + 0x12345) & 0xFFFFFFFF
    s ^= s >> 16
    return (s * 0x45679) & 0xFFFFFFFF
