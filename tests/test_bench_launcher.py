"""bench.py --gpus N must be runnable exactly as the driver runs it (`python3 bench.py --gpus N`): the parent process starts
the N ranks as a child torch.distributed.run and relays rank 0's line; it never initialises the GPU and never replaces
itself.  CPU only: --dry-run stops before anything is launched; a fake rank script stands in for the real ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e, timeout=300)


def test_dry_run_prints_the_launch_command():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = d["launch"]
    assert d["n_gpus"] == 2
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]       # the ranks get the same arguments, minus --dry-run


def test_single_gpu_does_not_launch():
    r = _run(["--gpus", "1", "--dry-run"])
    assert r.returncode == 0 and json.loads(r.stdout)["launch"] is None


def test_rank_refuses_a_world_that_disagrees_with_gpus():
    # started as a rank (WORLD_SIZE set) with another --gpus: an argument error, reported before any GPU work
    r = _run(["--gpus", "4"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_launcher_relays_rank0_line_and_exit_code(tmp_path):
    """The relay logic with stand-in ranks: python -m torch.distributed.run is replaced by a stub module on PYTHONPATH."""
    pkg = tmp_path / "torch" / "distributed"
    pkg.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "run.py").write_text(
        "import json, os, sys\n"
        "print('rank noise that is not the result line')\n"
        "if os.environ.get('FAKE_FAIL'): sys.exit(3)\n"
        "print(json.dumps({'metric': 'm', 'value': 1.0, 'n_gpus': int(sys.argv[sys.argv.index('--gpus') + 1])}))\n")
    env = {"PYTHONPATH": str(tmp_path)}
    r = _run(["--gpus", "2", "--steps", "1"], env=env)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2          # exactly one JSON line on stdout
    assert "rank noise" in r.stderr
    r = _run(["--gpus", "2", "--steps", "1"], env={**env, "FAKE_FAIL": "1"})
    assert r.returncode == 3 and r.stdout.strip() == ""
