#!/usr/bin/env python3
"""Soak run under STEADY Lees-Edwards shear (developer tool): the box tilt follows the wrapped strain and flips from +0.5 to -0.5
several times (PSEv1/VariantShearFunction.cc:34-43), soft-repulsive spheres, Brownian motion.  Fails on anything non-finite, on a
particle outside the sheared cell, or on an overlap the repulsion should have prevented.
  python3 tools/soak_shear.py [--n 50000] [--steps 2000] [--rate 2.0] [--error 1e-3]"""
import argparse, math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=50000); ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--rate", type=float, default=2.0); ap.add_argument("--phi", type=float, default=0.2)
    ap.add_argument("--error", type=float, default=1e-3)
    a = ap.parse_args()
    import torch
    from pse_amd import integrate, shear_function, variant, forces
    from pse_amd.system import System
    rng = np.random.default_rng(5)
    n, dt = a.n, 1e-3
    L = (4 * math.pi * n / (3 * a.phi)) ** (1 / 3)
    pos = rng.uniform(-L / 2, L / 2, size=(n, 3))
    s = System(pos, (L, L, L, 0.0), dt=dt)
    ff = shear_function.steady(dt=dt, shear_rate=a.rate)
    s.box_tilt_variant = variant.shear_variant(ff, a.steps, max_strain=0.5)
    pse = integrate.PSEv1(group=s.all(), T=1.0, seed=11, xi=0.5, error=a.error, function_form=ff)
    forces.HarmonicRepulsion(pse, k=200.0, sigma=2.0)
    t0 = time.time()
    flips, last = 0, 0.0
    for blk in range(a.steps // 100):
        for _ in range(100):
            s.run(1)
            if s.box[3] < last - 0.5:
                flips += 1
            last = s.box[3]
        p = s.pos[:, :3]
        assert bool(torch.isfinite(p).all()), "non-finite positions"
        fx = (p[:, 0] - s.box[3] * p[:, 1]) / L
        inside = max(float(fx.abs().max()), float((p[:, 1] / L).abs().max()), float((p[:, 2] / L).abs().max()))
        assert inside <= 0.5 + 1e-9, inside
        print(blk, "xy", round(s.box[3], 4), "flips", flips, "max|frac|", round(inside, 6), "m", pse.cpp_method.lanczosIterations(),
              "maxF", round(float(s.net_force[:, :3].abs().max()), 2), flush=True)
    torch.cuda.synchronize()
    assert flips >= int(a.rate * a.steps * dt - 0.5), flips
    print("%d steps in %.2f s, %d tilt flips: ok" % (a.steps, time.time() - t0, flips))


if __name__ == "__main__":
    main()
