"""Generates tests/golden/pse_oracle_golden.json from the oracle (run once, after the oracle was pinned to the
KATs of SURVEY.md 8c).  The reference has no fixtures of its own and cannot be built or imported here."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from conftest import make_suspension  # noqa: E402
from oracle import pse_port as pp  # noqa: E402

pos, force, box = make_suspension(64, L=16.0, xy=0.2, seed=2024, fseed=2025)
xi, error, kT, dt, seed, ts = 0.5, 1e-3, 1.0, 1e-3, 31337, 42
p = pp.select_params(box, xi, error, 0.5)
ub, m = pp.brownian_velocity(pos, force, box, p, kT, dt, seed, ts)
out = {
    "pos": pos.tolist(), "force": force.tolist(), "box": list(box), "xi": xi, "error": error, "kT": kT, "dt": dt,
    "seed": seed, "timestep": ts, "u_direct": pp.mobility_direct(pos, force, box, xi).tolist(),
    "u_brownian_port": ub.tolist(), "lanczos_m": int(m),
    "philox_1_2_3_4_5_6": [int(x) for x in pp.philox4x32(1, 2, 3, 4, 5, 6)], "hash_seed_1": pp.hash_seed(1),
    "params": {k: (list(v) if isinstance(v, tuple) else v) for k, v in p.items()},
}
json.dump(out, open(os.path.join(os.path.dirname(__file__), "pse_oracle_golden.json"), "w"))
print("written", len(json.dumps(out)), "bytes")
