#!/usr/bin/env python3
"""Soak run (developer tool): a few thousand Brownian steps of a small suspension with the kept neighbour list in use; prints the
mean-square displacement against 6 D0 t, the Lanczos counts and the list statistics, and fails on anything non-finite.
  python3 tools/soak.py [--n 20000] [--steps 3000] [--dt 2e-4]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20000); ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--dt", type=float, default=2e-4); ap.add_argument("--phi", type=float, default=0.2)
    a = ap.parse_args()
    import torch
    from conftest import make_suspension, to4
    import pse_amd
    n = a.n
    pos, force, box = make_suspension(n, phi=a.phi)
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=7)
    dpos = to4(pos, 1.0); vel = to4(np.zeros((n, 3)), 1.0); dF = to4(0.0 * force)
    accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    L = np.array(box[:3])
    m, ms = 2, []
    for ts in range(a.steps):
        m = eng.step(dpos, vel, accel, image, dF, 1.0, a.dt, ts, lanczos_m=m)
        ms.append(m)
        if (ts + 1) % 500 == 0:
            p = dpos.cpu().numpy()[:, :3] + image.cpu().numpy() * L
            assert np.isfinite(p).all(), "non-finite positions"
            msd = ((p - pos) ** 2).sum(1).mean()
            print(f"step {ts + 1}: MSD {msd:.4f} vs free 6 D0 t = {6 * a.dt * (ts + 1):.4f}  m in [{min(ms)}, {max(ms)}]  nlist {eng.neighbor_stats()}", flush=True)
    st = eng.neighbor_stats()
    assert st[1] + st[2] == a.steps
    print("ok")

if __name__ == "__main__":
    main()
