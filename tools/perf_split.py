#!/usr/bin/env python3
"""The two halves of the two-rank FUNCTIONAL split (pse_brownian_velocity_part) timed one after the other on ONE GPU: rank 0's chain
(sort + near field + Lanczos), rank 1's chain (sort + spread + transforms + gather), the integration both ranks do.  On two GPUs the
halves run side by side and meet in one all-reduce of 32 N bytes per direction: step = max(half) + all-reduce + integrate (DESIGN.md
section 6 prices the link).  python3 tools/perf_split.py [--n 1000000] [--phi 0.1] [--grid 256] [--steps 20]"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--phi", type=float, default=0.1)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    import torch
    import pse_amd
    from conftest import make_suspension, to4
    import numpy as np
    pos, force, box = make_suspension(a.n, phi=a.phi)
    xi = math.pi * a.grid / (2.0 * box[0] * math.sqrt(-math.log(1e-3)))
    eng = pse_amd.Engine(a.n, box, xi=xi, error=1e-3, seed=1, grid=(a.grid,) * 3)
    dpos, dF, vel = to4(pos), to4(force), to4(np.zeros((a.n, 3)), 1.0)
    accel = torch.zeros((a.n, 3), dtype=torch.float64, device="cuda"); image = torch.zeros((a.n, 3), dtype=torch.int32, device="cuda")
    rng = np.random.default_rng(3)
    moved = [dpos.clone() for _ in range(4)]
    for p in moved:      # (positions displaced beyond r_buff / 2 from call to call: every call sorts and walks the cells, as a step does)
        p[:, :3] += torch.tensor(rng.uniform(-0.5, 0.5, (a.n, 3)), dtype=torch.float64, device="cuda")
    m = 2
    for k in range(4):
        _, m = eng.brownian_velocity_part(moved[k % 4], dF, 1.0, 1e-3, k, 1, vel=vel, lanczos_m=m)
        eng.brownian_velocity_part(moved[k % 4], dF, 1.0, 1e-3, k, 2, vel=vel)
    res = {}
    for parts, name in ((1, "real-space half (rank 0: sort + near field + Lanczos, m = %d)" % m), (2, "wave-space half (rank 1: sort + far field + k-space noise)"), (3, "whole evaluation (one GPU)")):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(a.steps):
            eng.brownian_velocity_part(moved[k % 4], dF, 1.0, 1e-3, 10 + k, parts, vel=vel, lanczos_m=m)
        torch.cuda.synchronize()
        res[parts] = (time.perf_counter() - t0) / a.steps * 1e3
        print(f"{name}: {res[parts]:.3f} ms")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(a.steps):
        eng.integrate(dpos, vel, accel, image, dF, 1e-9)
    torch.cuda.synchronize()
    ti = (time.perf_counter() - t0) / a.steps * 1e3
    print(f"integration (both ranks): {ti:.3f} ms")
    for gbs in (50.0, 64.0, 100.0):
        tl = 32.0 * a.n / (gbs * 1e9) * 1e3
        print(f"two GPUs at {gbs:.0f} GB/s per direction: all-reduce of 32 N bytes each way {tl:.3f} ms -> step {max(res[1], res[2]) + tl + ti:.3f} ms "
              f"= {(res[3] + ti) / (max(res[1], res[2]) + tl + ti):.2f}x the single GPU's {res[3] + ti:.3f} ms")


if __name__ == "__main__":
    main()
