"""bench.py --gpus N must be runnable exactly as the driver runs it (`python3 bench.py --gpus N`): the parent process starts
the N ranks itself (one child per GPU with the environment torch.distributed.run would give them), relays rank 0's line and
the first failing exit code; it never initialises the GPU and never replaces itself.  CPU only: --dry-run stops before
anything is launched; a stand-in script plays the ranks."""
import argparse
import importlib.util
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
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--n", "1000", "--dry-run"])
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = d["launch"]
    assert d["n_gpus"] == 2
    assert cmd[1] == BENCH
    assert cmd[2:] == ["--gpus", "2", "--steps", "3", "--warmup", "1", "--n", "1000"]   # the ranks get the same arguments, minus --dry-run
    assert d["env"]["MASTER_ADDR"] == "127.0.0.1" and int(d["env"]["MASTER_PORT"]) > 0 and d["env"]["WORLD_SIZE"] == "2"


def test_single_gpu_does_not_launch():
    r = _run(["--gpus", "1", "--dry-run"])
    assert r.returncode == 0 and json.loads(r.stdout)["launch"] is None


def test_rank_refuses_a_world_that_disagrees_with_gpus():
    # started as a rank (WORLD_SIZE set) with another --gpus: an argument error, reported before any GPU work
    r = _run(["--gpus", "4"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


STAND_IN = """
import json, os, sys, time
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
assert os.environ['LOCAL_RANK'] == os.environ['RANK'] and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0
print(f'noise from rank {rank}')
mode = os.environ.get('FAKE', '')
if mode == 'fail' and rank == 1:
    sys.exit(3)
if mode == 'fail':
    time.sleep(600)            # a rank waiting in a collective for the one that died: the launcher must end it
if rank == 0:
    print(json.dumps({'metric': 'm', 'value': 1.0, 'n_gpus': world, 'argv': sys.argv[1:]}))
"""


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_launcher_relays_rank0_line_and_exit_code(tmp_path, capfd, monkeypatch):
    script = tmp_path / "rank.py"
    script.write_text(STAND_IN)
    bench = _bench_module()
    args = argparse.Namespace(gpus=3, dry_run=False)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    rc = bench.launch_ranks(args, ["--gpus", "3", "--n", "1000", "--dry-run"], script=str(script))
    out, err = capfd.readouterr()
    assert rc == 0
    lines = out.strip().splitlines()
    d = json.loads(lines[-1])
    assert len(lines) == 1 and d["n_gpus"] == 3 and d["argv"] == ["--gpus", "3", "--n", "1000"]     # exactly one JSON line on stdout
    assert all(f"noise from rank {r}" in err for r in range(3))
    import time
    monkeypatch.setenv("FAKE", "fail")
    t0 = time.time()
    rc = bench.launch_ranks(args, ["--gpus", "3"], script=str(script))
    out, err = capfd.readouterr()
    assert rc == 3 and out.strip() == "" and time.time() - t0 < 60
