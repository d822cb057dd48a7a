#!/usr/bin/env python3
"""NumPy model of the wave-per-row z transforms (k_zfft_rows in csrc/pse_kernels.hip: NC = R0 x 8 x 8 at Nz = 256 / 512; k_zfft_rows_g
in csrc/pse_zfft.hip: NC = R0 x R1 x R2 at the other 2^a 3^b 5^c sizes): the index algebra of the three stages (lane l < L = NC / R0 holds
the points l + L r), the real <-> half-spectrum step around them, and the normalisation (unnormalised both ways, as rocFFT's real
transforms) -- checked against numpy.fft for every size the kernels are instantiated at."""
import numpy as np


def cfft(z, R0, R1, R2, inverse):
    NC = len(z)
    L = NC // R0
    sg = 1.0 if inverse else -1.0
    assert NC == R0 * R1 * R2 and L == R1 * R2 and L <= 64
    W = lambda n, e: np.exp(sg * 2j * np.pi * e / n)   # noqa: E731
    # stage 1: lane l, points l + L r -> k0, times W_NC^{l k0}
    a = np.zeros((R0, L), complex)
    for l in range(L):
        v = np.array([z[l + L * r] for r in range(R0)])
        for k0 in range(R0):
            a[k0, l] = sum(v[r] * W(R0, r * k0) for r in range(R0)) * W(NC, l * k0)
    # stage 2: l = lo + R2 s: DFT over s -> k1, times W_L^{lo k1}
    b = np.zeros((R0, R1, R2), complex)   # [k0][k1][lo]
    for k0 in range(R0):
        for lo in range(R2):
            v = np.array([a[k0, lo + R2 * s] for s in range(R1)])
            for k1 in range(R1):
                b[k0, k1, lo] = sum(v[s] * W(R1, s * k1) for s in range(R1)) * W(L, lo * k1)
    # stage 3: DFT over lo -> k2: Z[k0 + R0 k1 + R0 R1 k2]
    Z = np.zeros(NC, complex)
    for k0 in range(R0):
        for k1 in range(R1):
            for k2 in range(R2):
                Z[k0 + R0 * k1 + R0 * R1 * k2] = sum(b[k0, k1, lo] * W(R2, lo * k2) for lo in range(R2))
    return Z


def r2c(x, f):
    N = len(x); NC = N // 2
    z = x[0::2] + 1j * x[1::2]
    Z = cfft(z, *f, False)
    X = np.zeros(NC + 1, complex)
    for k in range(NC + 1):
        zk, zc = Z[k % NC], np.conj(Z[(NC - k) % NC])
        X[k] = 0.5 * (zk + zc) - 0.5j * np.exp(-2j * np.pi * k / N) * (zk - zc)
    return X


def c2r(X, f):
    NC = len(X) - 1; N = 2 * NC
    Z = np.zeros(NC, complex)
    for k in range(NC):
        xk, xc = X[k], np.conj(X[NC - k])
        Z[k] = (xk + xc) + 1j * np.exp(2j * np.pi * k / N) * (xk - xc)
    z = cfft(Z, *f, True)
    x = np.zeros(N)
    x[0::2] = z.real; x[1::2] = z.imag
    return x


# Nz -> (R0, R1, R2) of the instantiations (pse_zfft.hip ZFFT_SIZES; 256 / 512: pse_kernels.hip)
SIZES = {256: (2, 8, 8), 512: (4, 8, 8), 360: (3, 6, 10), 270: (3, 5, 9), 180: (2, 5, 9), 240: (2, 6, 10), 300: (3, 5, 10), 320: (4, 4, 10),
         384: (3, 8, 8), 400: (4, 5, 10), 450: (5, 5, 9), 480: (4, 6, 10), 500: (5, 5, 10)}

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for N, f in SIZES.items():
        x = rng.normal(size=N)
        X = r2c(x, f)
        y = c2r(np.fft.rfft(x), f)
        print(N, f, "r2c %.2e" % np.abs(X - np.fft.rfft(x)).max(), "c2r (unnormalised: N x) %.2e" % np.abs(y - N * x).max())
