import sys, os
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import to4
import importlib.util
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests/test_gpu_random_configs.py")); T = importlib.util.module_from_spec(spec); spec.loader.exec_module(T)
from oracle import pse_port as o
import pse_amd
o.lib()

def rel(a, b): return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)

def run(tag, box, xi, err, grid, pos, force):
    p = o.select_params(box, xi, err, 0.5, grid=grid)
    eng = pse_amd.Engine(max(len(pos), 8), box, xi=xi, error=err, grid=grid or (0, 0, 0), seed=1)
    g = eng.debug_spread(to4(pos), to4(force))
    gref = np.asarray(o.spread(pos, force, box, p))
    uw = eng.mobility(to4(pos), to4(force), parts=2).cpu().numpy()[:, :3]
    ref = o.mobility_wave(pos, force, box, p)
    import ctypes
    i = eng.info()
    ug = np.zeros((3, i["Nx"], i["Ny"], i["Nz"]))
    pse_amd._lib.check(eng._lib.pse_debug_copy_grid(eng._h, 1, ug.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
    fh = np.fft.rfftn(gref, axes=(1, 2, 3))
    ugref = np.fft.irfftn(o.wave_scale(fh, box, p), s=p["grid"], axes=(1, 2, 3), norm="forward")
    d = np.abs(ug - ugref)
    print("   grid after inverse FFT err", d.max() / np.abs(ugref).max(), "worst node", np.unravel_index(d.argmax(), d.shape), "eta", p["eta"])
    # spectrum of the difference: which wave-vector indices carry it
    dh = np.abs(np.fft.fftn(ug - ugref, axes=(1, 2, 3))).sum(0)
    idx = np.argsort(dh.ravel())[::-1][:6]
    print("   worst modes", [tuple(int(v) for v in np.unravel_index(q, dh.shape)) for q in idx], [float(dh.ravel()[q]) for q in idx][:3])
    print(tag, "n", len(pos), "grid", p["grid"], "P", p["P"], "xy", round(box[3], 2), "spread err", np.abs(g - gref).max() / np.abs(gref).max(), "wave err", rel(uw, ref), flush=True)
    eng.close()

for s in (2, 5):
    c = T.config(s)
    run(f"seed{s}", c["box"], c["xi"], c["err"], c["grid"], c["pos"], c["force"])
    b = c["box"]
    # no tilt
    pos0 = c["pos"].copy(); pos0[:, 0] -= b[3] * pos0[:, 1]
    run(f"seed{s} xy=0", (b[0], b[1], b[2], 0.0), c["xi"], c["err"], c["grid"], pos0, c["force"])
    # more particles in the same box
    rng = np.random.default_rng(5); n = 1500
    f = rng.uniform(-0.5, 0.5, (n, 3)); pos = np.empty((n, 3)); pos[:, 1] = f[:, 1] * b[1]; pos[:, 2] = f[:, 2] * b[2]; pos[:, 0] = f[:, 0] * b[0] + b[3] * pos[:, 1]
    run(f"seed{s} n=1500", b, c["xi"], c["err"], c["grid"], pos, rng.normal(size=(n, 3)))
    # one particle at a time
    for k in range(0, 1):
        run(f"seed{s} only particle {k}", b, c["xi"], c["err"], c["grid"], c["pos"][k:k + 1], c["force"][k:k + 1])
    # cubic grid of the largest size
    gmax = max(o.select_params(b, c["xi"], c["err"], 0.5, grid=c["grid"])["grid"])
    run(f"seed{s} cubic grid", b, c["xi"], c["err"], (gmax,) * 3, c["pos"], c["force"])
