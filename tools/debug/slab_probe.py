#!/usr/bin/env python3
"""Loopback slab team against the single-GPU engine on odd geometries: prints the relative difference of M.F per rank.
  python3 tools/debug/slab_probe.py nx:ny:nz:world:P ..."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PSE_WAVE_MODE", "slab")


def main():
    from conftest import make_suspension, to4
    import pse_amd
    from pse_amd.sharded import LoopbackSimulation
    for spec in sys.argv[1:]:
        nx, ny, nz, world, P = (int(v) for v in spec.split(":"))
        n = 2500
        pos, force, box = make_suspension(n, L=24.0, xy=0.1)
        Lx = nx * 24.0 / 256.0 if nx > 256 else 24.0
        pos[:, 0] = (pos[:, 0] - 0.1 * pos[:, 1]) * (Lx / 24.0) + 0.1 * pos[:, 1]
        box = (Lx, 24.0, 24.0, 0.1)
        kw = dict(xi=0.5, error=1e-3, seed=12, grid=(nx, ny, nz), P=P)
        try:
            ref = pse_amd.Engine(n, box, **kw)
            sim = LoopbackSimulation(n, box, world, **kw)
            sim.load(pos, force)
            out = []
            for parts in (2, 1, 3):
                u_ref = ref.mobility(to4(pos), to4(force), parts=parts).cpu().numpy()[:, :3]
                vels = sim.mobility(parts=parts)
                out.append(max(np.linalg.norm(v.cpu().numpy()[:, :3] - u_ref) / np.linalg.norm(u_ref) for v in vels))
            print(spec, "wave %.2e real %.2e both %.2e" % tuple(out), flush=True)
        except Exception as e:
            print(spec, "error", str(e)[:200], flush=True)


if __name__ == "__main__":
    main()
