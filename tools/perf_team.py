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
    ap.add_argument("--solo", type=int, default=-1, help="after the full calls: time Brownian evaluations that queue the work of this "
                    "rank only (both lanes, its share of the copies): one rank's critical path with the GPU to itself")
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
    if a.solo >= 0:
        # a full Brownian evaluation at fixed positions leaves every member's buffers in place; then only rank `solo` works
        vels, m = sim.brownian_velocity(1.0, 1e-3, 50, lanczos_m=m)
        sim.team.debug_solo(a.solo)
        for it in range(3):
            sim.brownian_velocity(1.0, 1e-3, 50, lanczos_m=m)
        torch.cuda.synchronize()
        ts = []
        for it in range(max(a.steps, 20)):
            t0 = time.perf_counter()
            sim.brownian_velocity(1.0, 1e-3, 50, lanczos_m=m)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        i = sim.engines[a.solo].info()
        print(f"solo rank {a.solo} of {a.ranks}: Brownian evaluation (no Euler update) {ts[len(ts) // 2] * 1e3:.3f} ms median, "
              f"{ts[0] * 1e3:.3f} min, {ts[-1] * 1e3:.3f} max over {len(ts)} calls; m = {m}, exchanges = {i['lanczos_exchanges']}, "
              f"mat-vecs = {i['lanczos_matvecs']}")
        sim.team.debug_solo(-1)
        return
    for e in sim.engines[:1]:
        e.set_timing(True)
    sim.engines[0].set_timing(True)
    m = sim.step(1.0, 1e-3, 99, lanczos_m=m)
    i = sim.engines[0].info()
    print("rank 0 phases ms:", {k: round(v, 4) for k, v in i.items() if k.startswith("t_") and v > 0})
    print(f"Lanczos: m = {i['lanczos_m']}, near-field mat-vecs = {i['lanczos_matvecs']}, exchanges = {i['lanczos_exchanges']}")


if __name__ == "__main__":
    main()
