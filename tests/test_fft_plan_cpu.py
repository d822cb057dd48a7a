"""Host-side mirrors of the index algebra of the register x / y passes (pse_amd/csrc/pse_kernels.hip: k_xfft_scale_cols, k_yfft_regs,
dft_small<6>, dft_small<10>), in numpy: the three-stage decimation in frequency that leaves X[k0 + R0 k1 + R0 R1 k2] in "register k2 of
lane (k0, k1)", its inverse from that digit-reversed order, the Good-Thomas maps of the radix-6 / radix-10 butterflies, the
natural-order hand-over of the y pass, and the LDS paddings the kernels are instantiated with (bank model of
tools/debug/lds_banks_xcols.py, lane groups from MI355X_MICROARCH.md).  No GPU: the kernels themselves are held to the port in
tests/test_gpu_parity.py::test_fused_x_pass_matches_port."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "debug"))

PLANS = [(512, 8, 8), (360, 10, 6), (256, 8, 8), (256, 4, 8)]           # N, R0, R1 as instantiated (launch_xfft_scale, launch_yfft)


def forward_stages(x, R0, R1):
    """x[N] -> v[k0, k1, k2] = X[k0 + R0 k1 + R0 R1 k2], stage by stage as the kernel does"""
    N = len(x)
    M1 = N // R0
    R2 = M1 // R1
    W = lambda n, e: np.exp(-2j * np.pi * e / n)            # noqa: E731
    # stage 1: thread n holds x[n + M1 r]; radix R0 over r, times W_N^{n k0}
    B = np.empty((R0, M1), complex)
    for n in range(M1):
        a = x[n + M1 * np.arange(R0)]
        B[:, n] = np.fft.fft(a) * W(N, n * np.arange(R0))
    # stage 2: lane (k0, n'') holds B[k0][R2 s + n'']; radix R1 over s, times W_M1^{n'' k1}
    C = np.empty((R0, R1, R2), complex)
    for k0 in range(R0):
        for nn in range(R2):
            b = B[k0, R2 * np.arange(R1) + nn]
            C[k0, :, nn] = np.fft.fft(b) * W(M1, nn * np.arange(R1))
    # stage 3: lane (k0, k1) holds C[k0][k1][n'']; radix R2 over n''
    return np.fft.fft(C, axis=2)


def inverse_stages(v, R0, R1):
    """the same stages backwards from the digit-reversed order (unnormalised)"""
    R2 = v.shape[2]
    M1 = R1 * R2
    N = R0 * M1
    Wc = lambda n, e: np.exp(2j * np.pi * e / n)            # noqa: E731
    C = np.fft.ifft(v, axis=2) * R2                          # over k2 -> n''
    B = np.empty((R0, M1), complex)
    for k0 in range(R0):
        for nn in range(R2):
            c = C[k0, :, nn] * Wc(M1, nn * np.arange(R1))   # times conj W_M1^{n'' k1}, then over k1 -> s
            B[k0, R2 * np.arange(R1) + nn] = np.fft.ifft(c) * R1
    x = np.empty(N, complex)
    for n in range(M1):
        b = B[:, n] * Wc(N, n * np.arange(R0))              # times conj W_N^{n k0}, then over k0 -> r
        x[n + M1 * np.arange(R0)] = np.fft.ifft(b) * R0
    return x


@pytest.mark.parametrize("N,R0,R1", PLANS)
def test_three_stage_transform_and_its_inverse(N, R0, R1):
    rng = np.random.default_rng(N)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    v = forward_stages(x, R0, R1)
    X = np.fft.fft(x)
    R2 = N // (R0 * R1)
    k0, k1, k2 = np.meshgrid(np.arange(R0), np.arange(R1), np.arange(R2), indexing="ij")
    assert np.abs(v - X[k0 + R0 * k1 + R0 * R1 * k2]).max() < 1e-11 * np.abs(X).max()
    assert np.abs(inverse_stages(v, R0, R1) - N * x).max() < 1e-11 * N


@pytest.mark.parametrize("R", [6, 10])
def test_good_thomas_butterflies(R):
    """dft_small<6>, dft_small<10>: x[(Q n1 + 2 n2) mod R] -> X[(Q k1 + (Q + 1) k2) mod R], Q = R / 2, no twiddles"""
    Q = R // 2
    rng = np.random.default_rng(R)
    x = rng.standard_normal(R) + 1j * rng.standard_normal(R)
    sm = np.array([x[(2 * n2) % R] + x[(Q + 2 * n2) % R] for n2 in range(Q)])
    df = np.array([x[(2 * n2) % R] - x[(Q + 2 * n2) % R] for n2 in range(Q)])
    S, D = np.fft.fft(sm), np.fft.fft(df)
    out = np.empty(R, complex)
    for k2 in range(Q):
        out[((Q + 1) * k2) % R] = S[k2]
        out[(Q + (Q + 1) * k2) % R] = D[k2]
    assert np.abs(out - np.fft.fft(x)).max() < 1e-13
    assert len({((Q + 1) * k2) % R for k2 in range(Q)} | {(Q + (Q + 1) * k2) % R for k2 in range(Q)}) == R   # every output written once


@pytest.mark.parametrize("N", [256, 512])
def test_natural_order_handover_of_the_y_pass(N):
    """k_yfft_regs: lane (k0, k1), register k2 parks ky = k0 + 8 k1 + 64 k2 at k0 + 9 k1 + 72 k2 = ky + ky // 8; thread n of layout A
    reads y = n + M1 r at n + n // 8 + (M1 + M1 // 8) r -- the same positions, all distinct, inside the column"""
    R0 = R1 = 8
    M1 = N // R0
    R2 = M1 // R1
    pos = {}
    for k0 in range(R0):
        for k1 in range(R1):
            for k2 in range(R2):
                ky = k0 + 8 * k1 + 64 * k2
                assert k0 + 9 * k1 + 72 * k2 == ky + ky // 8
                pos[ky] = ky + ky // 8
    assert sorted(pos) == list(range(N)) and len(set(pos.values())) == N and max(pos.values()) < N + N // 8
    for n in range(M1):
        for r in range(R0):
            assert n + n // 8 + (M1 + M1 // 8) * r == pos[n + M1 * r]


# (N, R0, R1, KB, P0, P1, CS, columns per wave) of every instantiation (256 = 8 x 8 x 4: the y pass)
PADDINGS = [(512, 8, 8, 4, 72, 9, 578, 1), (360, 10, 6, 4, 54, 9, 538, 1), (256, 8, 8, 8, 44, 5, 359, 1), (256, 4, 8, 8, 72, 9, 295, 2)]


@pytest.mark.parametrize("N,R0,R1,KB,P0,P1,CS,CPW", PADDINGS)
def test_lds_paddings_are_conflict_free_in_the_bank_model(N, R0, R1, KB, P0, P1, CS, CPW):
    import lds_banks_xcols as banks
    M1 = N // R0
    R2 = M1 // R1
    # positions of a column stay inside it and do not collide
    b = {P0 * k0 + n for k0 in range(R0) for n in range(M1)}
    c = {P0 * k0 + P1 * k1 + n for k0 in range(R0) for k1 in range(R1) for n in range(R2)}
    assert len(b) == R0 * M1 and len(c) == N and max(b | c) < CS and (64 // CPW) * R2 <= CS
    w = banks.evaluate(N, R0, R1, KB, P0, P1, CS, CPW)
    for name, cost in w.items():
        # every access one LDS pass per lane group, but the layout A read of the inverse (two) and, at R2 = 4, the read of the
        # inner exchange (two: four-point rows are 64 bytes)
        limit = 2.0 if name == "E1' read (A)" or (R2 == 4 and name == "E2 read") else 1.0
        assert cost <= limit, (name, cost)
