#!/usr/bin/env python3
"""Per-rank cost of the slab-decomposed step from an in-process team (all G ranks on one GPU, copies instead of RCCL): the
kernels of the G ranks run one after another, so (time of a team step) / G is the per-rank COMPUTE time of a G-GPU run; link
time is added on paper.  python3 tools/perf_team.py [--ranks 8] [--n 1000000] [--grid 256] [--steps 5]"""
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
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--phi", type=float, default=0.1)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    import torch
    from conftest import make_suspension
    from pse_amd.sharded import LoopbackSimulation
    pos, force, box = make_suspension(a.n, phi=a.phi)
    xi = math.pi * a.grid / (2.0 * box[0] * math.sqrt(-math.log(1e-3)))
    sim = LoopbackSimulation(a.n, box, a.ranks, xi=xi, error=1e-3, seed=1, grid=(a.grid,) * 3)
    sim.load(pos, force)
    m = 2
    for it in range(3):
        m = sim.step(1.0, 1e-3, it, lanczos_m=m)
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(a.steps):
        m = sim.step(1.0, 1e-3, 10 + it, lanczos_m=m)
    torch.cuda.synchronize(); t = (time.time() - t0) / a.steps
    print(f"team of {a.ranks}: {t * 1e3:.3f} ms per team step -> {t * 1e3 / a.ranks:.3f} ms per rank (compute only), m={m}")
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(a.steps):
        sim.mobility()
    torch.cuda.synchronize(); t = (time.time() - t0) / a.steps
    print(f"team of {a.ranks}: M.F {t * 1e3:.3f} ms per team eval -> {t * 1e3 / a.ranks:.3f} ms per rank")
    for e in sim.engines[:1]:
        e.set_timing(True)
    sim.engines[0].set_timing(True)
    m = sim.step(1.0, 1e-3, 99, lanczos_m=m)
    i = sim.engines[0].info()
    print("rank 0 phases ms:", {k: round(v, 4) for k, v in i.items() if k.startswith("t_") and v > 0})


if __name__ == "__main__":
    main()
