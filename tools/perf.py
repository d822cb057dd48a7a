#!/usr/bin/env python3
"""Per-phase timings of the engine on a synthetic suspension (developer tool; bench.py is the contract).

  python3 tools/perf.py --n 1000000 --phi 0.1 --grid 256 [--error 1e-3] [--xy 0.0] [--steps 5] [--no-step]
Prints the hipEvent phase times of pse_mobility and pse_brownian_velocity, then untimed-loop rates.
"""
import argparse
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--phi", type=float, default=0.1)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--error", type=float, default=1e-3)
    ap.add_argument("--xi", type=float, default=0.0, help="0: xi from the grid (SURVEY 8d); with --grid 0 the reference rule picks the grid")
    ap.add_argument("--xy", type=float, default=0.0)
    ap.add_argument("--kT", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--no-step", action="store_true")
    ap.add_argument("--only-mf", action="store_true")
    a = ap.parse_args()

    import torch
    from conftest import make_suspension, to4
    import pse_amd

    n = a.n
    pos, force, box = make_suspension(n, phi=a.phi)
    L = box[0]
    box = (L, L, L, a.xy)
    grid = (a.grid,) * 3 if a.grid else (0, 0, 0)
    xi = a.xi if a.xi > 0 else math.pi * a.grid / (2 * L * math.sqrt(-math.log(a.error)))
    t0 = time.time()
    eng = pse_amd.Engine(n, box, xi=xi, error=a.error, seed=1, grid=grid)
    i = eng.info()
    print("create %.2fs" % (time.time() - t0), {k: i[k] for k in ("Nx", "Ny", "Nz", "P", "rcut", "eta", "ncell_x", "device_bytes")})
    dpos, dF = to4(pos, 1.0), to4(force)
    vel = to4(np.zeros((n, 3)), 1.0)
    eng.set_timing(True)
    for it in range(3):
        eng.mobility(dpos, dF, vel=vel)
    i = eng.info()
    print("M.F phases ms:", {k: round(v, 4) for k, v in i.items() if k.startswith("t_") and v > 0})
    m = 2
    if not a.only_mf:
        for it in range(3):
            _, m = eng.brownian_velocity(dpos, dF, a.kT, 1e-3, it, vel=vel, lanczos_m=m)
        i = eng.info()
        print("Brownian phases ms:", {k: round(v, 4) for k, v in i.items() if k.startswith("t_") and v > 0}, "m", m, i["lanczos_matvecs"])
    eng.set_timing(False)
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(a.steps):
        eng.mobility(dpos, dF, vel=vel)
    torch.cuda.synchronize(); t = (time.time() - t0) / a.steps
    print("M.F  %.3f ms/eval  -> %.1f evals/s" % (t * 1e3, 1 / t))
    if a.no_step or a.only_mf:
        return
    accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(a.steps):
        m = eng.step(dpos, vel, accel, image, dF, a.kT, 1e-3, 100 + it, lanczos_m=m)
    torch.cuda.synchronize(); t = (time.time() - t0) / a.steps
    print("step %.3f ms/step -> %.1f steps/s, %.3g particle-steps/s, m=%d" % (t * 1e3, 1 / t, n / t, m))
    # queue-only steps (pse_set_async: the Lanczos decision on the device, no read-back), eager and as a replayed hipGraph
    eng.set_async(True)
    for extra_note in ("eager",):
        for it in range(3):
            eng.step(dpos, vel, accel, image, dF, a.kT, 1e-3, 200 + it, lanczos_m=m)
        torch.cuda.synchronize(); t0 = time.time()
        for it in range(a.steps):
            eng.step(dpos, vel, accel, image, dF, a.kT, 1e-3, 300 + it, lanczos_m=m)
        torch.cuda.synchronize(); t = (time.time() - t0) / a.steps
        i = eng.info()
        print("queue-only step (%s) %.3f ms/step, m=%d status=%d" % (extra_note, t * 1e3, i["lanczos_m"], i["lanczos_status"]))
    s = torch.cuda.Stream()
    eng.set_stream(s.cuda_stream)
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    eng.set_timestep_offset(word)
    with torch.cuda.stream(s):
        eng.step(dpos, vel, accel, image, dF, a.kT, 1e-3, 400, lanczos_m=m)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        eng.step(dpos, vel, accel, image, dF, a.kT, 1e-3, 400, lanczos_m=m)
    for it in range(3):
        g.replay()
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(a.steps):
        g.replay()
    torch.cuda.synchronize(); t = (time.time() - t0) / a.steps
    i = eng.info()
    print("queue-only step (hipGraph replay) %.3f ms/step, m=%d status=%d" % (t * 1e3, i["lanczos_m"], i["lanczos_status"]))


if __name__ == "__main__":
    main()
