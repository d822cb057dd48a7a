#!/usr/bin/env python3
"""Time of the fused x pass per grid node as the x stride (Ny * Nzp * 16 B) changes: Nx x Ny x Nz grids on a box stretched alike.

  python3 tools/debug/xpass_probe.py 512:64:512 512:128:512 512:256:512 512:512:512
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from conftest import to4
    import pse_amd
    h = 24.0 / 256
    for spec in sys.argv[1:]:
        nx, ny, nz = (int(v) for v in spec.split(":"))
        box = (nx * h * 4, ny * h * 4, nz * h * 4, 0.0)
        n = 200_000
        rng = np.random.default_rng(1)
        pos = (rng.random((n, 3)) - 0.5) * np.array(box[:3])
        xi = math.pi / (2 * 4 * h * math.sqrt(-math.log(1e-3)))
        eng = pse_amd.Engine(n, box, xi=xi, error=1e-3, seed=1, grid=(nx, ny, nz))
        dpos, dF = to4(pos, 1.0), to4(rng.standard_normal((n, 3)))
        vel = to4(np.zeros((n, 3)), 1.0)
        eng.set_timing(True)
        ts = []
        for it in range(14):
            eng.mobility(dpos, dF, vel=vel)
            ts.append(eng.info()["t_scale"])
        i = eng.info()
        i["t_scale"] = min(ts[2:])
        med = sorted(ts[2:])[len(ts[2:]) // 2]
        nodes = nx * ny * (nz // 2 + 1)
        gb = 2 * 3 * 16 * nodes / 1e9
        m = 2
        for it in range(3):
            _, m = eng.brownian_velocity(dpos, dF, 1.0, 1e-3, it, vel=vel, lanczos_m=m)
        ib = eng.info()
        print(spec, "t_scale min %.4f median %.4f ms  %.2f TB/s  (fft %.3f / %.3f)   with noise %.4f ms" % (i["t_scale"], med, gb / i["t_scale"], i["t_fft_fwd"], i["t_fft_inv"], ib["t_scale"]), flush=True)
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
