"""Every developer switch the README names runs the parity cases of the code it selects, in a child process with the switch set
(the environment is read once per handle, in pse_create): no kernel family lives in the library without a test.
The cases are the ones of tests/test_gpu_parity.py, tests/test_gpu_slabs.py and tests/test_gpu_local.py that exercise the sizes
the switch matters at."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

X = "far_field_passes_at_the_switch_sizes"
CASES = [
    # (environment, files, -k expression)
    ({"PSE_OWN_Y": "0"}, ["test_gpu_parity.py"], X),                       # rocFFT's 2-D (y, z) plans everywhere
    ({"PSE_OWN_Y_POW2": "0"}, ["test_gpu_parity.py"], X),                  # ... at Ny = 256 / 512 only
    ({"PSE_OWN_Z": "0"}, ["test_gpu_parity.py"], X),                       # rocFFT's 1-D z transforms under the own y pass
    ({"PSE_XMIX": "1"}, ["test_gpu_parity.py"], X),                        # runtime radix plans where compile-time ones exist
    ({"PSE_NO_XFUSE": "1"}, ["test_gpu_parity.py"], X + " or wave_matches_port"),   # rocFFT 3-D + k_scale
    ({"PSE_XFFT_SMALL_KB": "8"}, ["test_gpu_parity.py"], X),               # eight-column x pass on small grids
    ({"PSE_YFFT_KB": "2"}, ["test_gpu_parity.py"], X),
    ({"PSE_YFFT_KB": "8"}, ["test_gpu_parity.py"], X),
    ({"PSE_OVERLAP": "0"}, ["test_gpu_parity.py"], "brownian_velocity_matches_port or step_integrates or wave_matches_port"),   # Brownian steps on one stream (the default of rounds 1-5)
    ({"PSE_OVERLAP": "-1"}, ["test_gpu_parity.py"], "brownian_velocity_matches_port or total_mobility"),
    ({"PSE_GATHER_BZ": "1"}, ["test_gpu_parity.py"], "wave_matches_port or both_halves"),
    ({"PSE_GATHER_BZ": "2"}, ["test_gpu_parity.py"], "wave_matches_port or both_halves"),
    ({"PSE_SPREAD_TZ": "8"}, ["test_gpu_parity.py"], "wave_matches_port or both_halves"),
    ({"PSE_SPREAD_TZ": "16"}, ["test_gpu_parity.py"], "wave_matches_port or both_halves"),
    ({"PSE_SPREAD_NW": "1"}, ["test_gpu_parity.py"], "wave_matches_port"),
    ({"PSE_SPREAD_NW": "4"}, ["test_gpu_parity.py"], "wave_matches_port"),
    ({"PSE_SIDE_PRIORITY": "low"}, ["test_gpu_parity.py"], "total_mobility or wave_matches_port"),
    ({"PSE_SIDE_PRIORITY": "default"}, ["test_gpu_local.py"], "velocities_match"),   # (an owned-particle rank's far-field lane is LOW by default)
    ({"PSE_SIDE_PRIORITY": "high"}, ["test_gpu_local.py"], "velocities_match"),
    ({"PSE_SKIN": "0"}, ["test_gpu_parity.py", "test_gpu_nlist.py"], "mreal_matches_oracle or brownian_velocity_matches_port or step_integrates"),
    ({"PSE_VQ": "0"}, ["test_gpu_parity.py"], "brownian_velocity_matches_port or step_integrates"),   # the mat-vec's neighbour rows as doubles
    ({"PSE_LANCZOS_EXTRA": "0"}, ["test_gpu_async.py"], "captured"),       # no gated iterations queued: the starting count must suffice
    ({"PSE_LANCZOS_EXTRA": "4"}, ["test_gpu_async.py", "test_gpu_local.py"], "captured or velocities_match"),
    ({"PSE_TEAM_SSTEP": "0"}, ["test_gpu_slabs.py"], "clustered or particle_group or follows_tilt"),  # one Lanczos iteration per exchange (replicated-state teams)
    ({"PSE_TEAM_LANES": "0"}, ["test_gpu_slabs.py", "test_gpu_local.py"], "team_of_eight or clustered or velocities_match"),   # one stream for everything
    ({"PSE_TEAM_SCHED": "0,0,0"}, ["test_gpu_slabs.py"], "team_of_eight or clustered"),
    ({"PSE_WAVE_MODE": "replicated"}, ["test_gpu_slabs.py"], "team_of_eight or clustered"),
    ({"PSE_WAVE_MODE": "slab"}, ["test_gpu_slabs.py"], "keeps_replicas or clustered"),
    ({"PSE_YSLAB_REGS": "0"}, ["test_gpu_local.py"], "velocities_match"),   # a slab rank's y pass by k_fft_cols also at Ny = 256
    # The cells are stored in blocks of b along z (x, z block, y, z in block; default b = 6 where an axis has at least twelve cells) so
    # that a wavefront's rows form a squat brick; PSE_CELL_BZ=0 is the plain (x, y, z) order.  Every near-field path (cell pass, pair
    # list, overflow rows, kept neighbour list, pair repulsion) must give the same answers in either (the test boxes have 6-8 cells
    # per axis: default = plain there, so b = 2 is what exercises the blocks).
    ({"PSE_CELL_BZ": "0"}, ["test_gpu_parity.py", "test_gpu_nlist.py"], _NEAR := "mreal_matches_oracle or pair_list_overflow or brownian_velocity_matches_port "
     "or step_integrates or pair_repulsion or reused_list or overflow_rows"),
    ({"PSE_CELL_BZ": "2"}, ["test_gpu_parity.py", "test_gpu_nlist.py"], _NEAR),
]


def _run_case(case):
    env, files, expr = case
    e = dict(os.environ, **env)
    # the children run ten at a time and the oracle they check against is OpenMP code: with the default (all cores each) ten of them
    # oversubscribe the host ten times over -- the far-field cases took 130-140 s each that way, 10-20 s with a share of the cores
    e["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 8) // 10))
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", *[os.path.join(ROOT, "tests", f) for f in files], "-k", expr],
                       env=e, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return r.returncode, r.stdout[-3000:], time.time() - t0


_results = {}


def _result(idx):
    """The child runs are independent of one another (each has its own process and handles): the first test that asks starts them
    all, ten at a time (the GPU box has 128 host cores; a child holds a few hundred MB of device memory), and every test reads its own
    outcome -- most of a child's time is interpreter and library start-up."""
    if not _results:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=10) as pool:
            for i, out in enumerate(pool.map(_run_case, CASES)):
                _results[i] = out
        try:     # (where the time of this module goes: one line per child, for whoever trims the suite)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "switch_children_seconds.txt"), "w") as f:
                for i, c in enumerate(CASES):
                    f.write(f"{_results[i][2]:7.1f} s  rc {_results[i][0]}  {c[0]}  {c[2]}\n")
        except OSError:
            pass
    return _results[idx]


@pytest.mark.parametrize("idx", range(len(CASES)), ids=[",".join(f"{k}={v}" for k, v in c[0].items()) for c in CASES])
def test_switch(idx):
    rc, tail, _seconds = _result(idx)
    assert rc == 0, tail
    assert " passed" in tail and "no tests ran" not in tail, tail      # the expression selected something
