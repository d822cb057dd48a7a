#!/usr/bin/env python3
"""NumPy model of the wave-per-row z transforms (k_zfft_rows in csrc/pse_kernels.hip): the index algebra of the three stages
(NC = R0 x 8 x 8 complex points, lane l holds points l + 64 r), the real <-> half-spectrum step around them, and the
normalisation (unnormalised both ways, as rocFFT's real transforms) -- checked against numpy.fft for Nz = 256 and 512."""
import numpy as np


def cfft(z, R0, inverse):
    NC = len(z)
    sg = 1.0 if inverse else -1.0
    assert NC == R0 * 64
    W = lambda n, e: np.exp(sg * 2j * np.pi * e / n)   # noqa: E731
    # stage 1: lane l, points l + 64 r -> k0, times W_NC^{l k0}
    a = np.zeros((R0, 64), complex)
    for l in range(64):
        v = np.array([z[l + 64 * r] for r in range(R0)])
        for k0 in range(R0):
            a[k0, l] = sum(v[r] * W(R0, r * k0) for r in range(R0)) * W(NC, l * k0)
    # stage 2: l = nn + 8 s: DFT over s -> k1, times W_64^{nn k1}
    b = np.zeros((R0, 8, 8), complex)   # [k0][k1][nn]
    for k0 in range(R0):
        for nn in range(8):
            v = np.array([a[k0, nn + 8 * s] for s in range(8)])
            for k1 in range(8):
                b[k0, k1, nn] = sum(v[s] * W(8, s * k1) for s in range(8)) * W(64, nn * k1)
    # stage 3: DFT over nn -> k2: Z[k0 + R0 k1 + 8 R0 k2]
    Z = np.zeros(NC, complex)
    for k0 in range(R0):
        for k1 in range(8):
            for k2 in range(8):
                Z[k0 + R0 * k1 + 8 * R0 * k2] = sum(b[k0, k1, nn] * W(8, nn * k2) for nn in range(8))
    return Z


def r2c(x, R0):
    N = len(x); NC = N // 2
    z = x[0::2] + 1j * x[1::2]
    Z = cfft(z, R0, False)
    X = np.zeros(NC + 1, complex)
    for k in range(NC + 1):
        zk, zc = Z[k % NC], np.conj(Z[(NC - k) % NC])
        X[k] = 0.5 * (zk + zc) - 0.5j * np.exp(-2j * np.pi * k / N) * (zk - zc)
    return X


def c2r(X, R0):
    NC = len(X) - 1; N = 2 * NC
    Z = np.zeros(NC, complex)
    for k in range(NC):
        xk, xc = X[k], np.conj(X[NC - k])
        Z[k] = (xk + xc) + 1j * np.exp(2j * np.pi * k / N) * (xk - xc)
    z = cfft(Z, R0, True)
    x = np.zeros(N)
    x[0::2] = z.real; x[1::2] = z.imag
    return x


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for N, R0 in ((256, 2), (512, 4)):
        x = rng.normal(size=N)
        X = r2c(x, R0)
        print(N, "r2c", np.abs(X - np.fft.rfft(x)).max())
        y = c2r(np.fft.rfft(x), R0)
        print(N, "c2r (unnormalised: N x)", np.abs(y - N * x).max())
