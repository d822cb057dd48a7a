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
    # the owned-particle run is a sequence of segments, each a fresh set of rank processes with a rendezvous port of its own
    # (split: the two-rank functional split, --gpus 2 only; host_fallback: only when no RCCL mode ended with a verified trajectory)
    assert d["segments"] == ["single", "one_stream", "lanes", "split", "host_fallback"]
    assert d["env"]["MASTER_ADDR"] == "127.0.0.1" and d["env"]["WORLD_SIZE"] == "2"
    r = _run(["--gpus", "2", "--replicated", "--dry-run"])           # one set of ranks: the launch line with its port
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert int(d["env"]["MASTER_PORT"]) > 0 and d["launch"][2:] == ["--gpus", "2", "--replicated"]


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


SEGMENT_STAND_IN = """
import json, os, sys, time
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
seg = sys.argv[sys.argv.index('--segment') + 1]
assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0 and 'TORCHELASTIC_USE_AGENT_STORE' not in os.environ
fake = os.environ.get('FAKE', '')
if seg == 'single':
    assert world == 1 and 'PSE_TEAM_LANES' not in os.environ
    print(json.dumps({'segment': seg, 'metric': 'm', 'ms_per_step': 8.0, 'config4_single_gpu': {'ms_per_step': 40.0}}))
    sys.exit(0)
if seg == 'split':
    assert world == 2 and 'PSE_TEAM_LANES' not in os.environ
    if rank == 0:
        print(json.dumps({'segment': seg, 'metric': 'm', 'value': 1e6 / 3.0, 'ms_per_step': 3.0, 'n_gpus': 2, 'verify': {'ok': True}, 'lanczos_status': 0,
                          'config4': {'ms_per_step': 12.0, 'verify': {'ok': True}}, 'config': {'parallelism': 'split'}}))
    sys.exit(0)
lanes = os.environ['PSE_TEAM_LANES']
assert lanes == ('1' if seg == 'lanes' else '0')
host = '--transport' in sys.argv and sys.argv[len(sys.argv) - 1 - sys.argv[::-1].index('--transport') + 1] == 'host'
if fake == 'rccl_down' and not host:
    sys.exit(5)                # (no communicator could be made)
if fake == 'lanes_hang' and seg == 'lanes':
    if rank == 1:
        sys.exit(7)
    time.sleep(600)
if fake == 'lanes_unverified' and seg == 'lanes':
    ok = False
else:
    ok = True
if rank == 0:
    ms = 2.0 if seg == 'lanes' else 2.5
    print('chatter')
    print(json.dumps({'segment': seg, 'metric': 'm', 'value': 1e6 / ms, 'ms_per_step': ms, 'n_gpus': world, 'verify': {'ok': ok}, 'lanczos_status': 0,
                      'config4': {'ms_per_step': 8.0 if seg == 'lanes' else 10.0, 'verify': {'ok': True}}, 'config': {'parallelism': 'lanes=' + lanes}}))
"""


def _supervise(tmp_path, capfd, monkeypatch, fake, gpus=3, extra=()):
    script = tmp_path / "seg.py"
    script.write_text(SEGMENT_STAND_IN)
    bench = _bench_module()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("FAKE", fake)
    monkeypatch.setenv("TORCHELASTIC_USE_AGENT_STORE", "True")           # must not reach the ranks of a segment
    ap = argparse.Namespace(gpus=gpus, dry_run=False, no_single=False, modes="both", no_verify=False, transport="rccl")
    for k in extra:
        setattr(ap, k, True)
    rc = bench.supervise_segments(ap, ["--gpus", str(gpus)], script=str(script))
    out, err = capfd.readouterr()
    return rc, out, err


def test_segments_both_modes_are_timed_and_the_faster_verified_one_is_the_value(tmp_path, capfd, monkeypatch):
    """`bench.py --gpus N`: a single-GPU segment, then the team in both lane modes, each a fresh set of rank processes; ONE line whose
    value is the faster verified mode, both modes recorded, the speed-ups against the single GPU measured in the same run."""
    rc, out, err = _supervise(tmp_path, capfd, monkeypatch, "")
    assert rc == 0, err
    lines = out.strip().splitlines()
    d = json.loads(lines[-1])
    assert len(lines) == 1 and "chatter" in err
    assert d["mode"] == "lanes" and d["ms_per_step"] == 2.0 and set(d["modes"]) == {"one_stream", "lanes"}
    assert d["modes"]["one_stream"]["ms_per_step"] == 2.5 and d["modes"]["lanes"]["verify"]["ok"]
    assert d["single_gpu"]["ms_per_step"] == 8.0 and abs(d["speedup_vs_single"] - 4.0) < 1e-12
    assert abs(d["config4"]["speedup_vs_single"] - 5.0) < 1e-12 and abs(d["modes"]["one_stream"]["config4"]["speedup_vs_single"] - 4.0) < 1e-12
    assert d["north_star"]["speedup_at_this_gpu_count"][1] == d["speedup_vs_single"]
    assert "segment" not in d and d["config"]["parallelism"] == "lanes=1"


def test_a_hang_in_the_second_mode_still_prints_the_first(tmp_path, capfd, monkeypatch):
    """Two lanes + a communication stream has never run over RCCL: a rank that dies there (the others then wait in a collective) ends
    that segment's children and the line of the one-stream mode is printed all the same."""
    import time
    t0 = time.time()
    rc, out, err = _supervise(tmp_path, capfd, monkeypatch, "lanes_hang")
    assert rc == 0 and time.time() - t0 < 90, err
    d = json.loads(out.strip().splitlines()[-1])
    assert d["mode"] == "one_stream" and d["ms_per_step"] == 2.5 and "error" in d["modes"]["lanes"] and "7" in d["modes"]["lanes"]["error"]


def test_no_rccl_mode_at_all_falls_back_to_the_host_transport_and_says_so(tmp_path, capfd, monkeypatch):
    """A team over RCCL has never run before the first multi-GPU node.  If neither lane mode ends with a verified trajectory the
    supervisor runs the one-stream step once more over the host transport: the line then still says whether the decomposition is
    right on these GPUs, labelled as the fallback it is; when an RCCL mode works the fallback is not run at all."""
    # (ranks that would share a GPU measure nothing about a multi-GPU node: with fewer GPUs than ranks -- none here -- there is no
    # fallback and the run ends with exit code 3 and no line; the variable is how this path is tested where GPUs are missing)
    monkeypatch.delenv("PSE_BENCH_FALLBACK_ANYWAY", raising=False)
    rc, out, err = _supervise(tmp_path, capfd, monkeypatch, "rccl_down")
    assert rc == 3 and out.strip() == "" and "no mode of the team finished" in err
    monkeypatch.setenv("PSE_BENCH_FALLBACK_ANYWAY", "1")
    rc, out, err = _supervise(tmp_path, capfd, monkeypatch, "rccl_down")
    assert rc == 0, err
    d = json.loads(out.strip().splitlines()[-1])
    assert d["mode"] == "host_fallback" and "fallback transport" in d["mode_note"] and d["ms_per_step"] == 2.5
    assert "error" in d["modes"]["one_stream"] and "error" in d["modes"]["lanes"] and d["modes"]["host_fallback"]["verify"]["ok"]
    rc, out, err = _supervise(tmp_path, capfd, monkeypatch, "")
    d = json.loads(out.strip().splitlines()[-1])
    assert rc == 0 and "host_fallback" not in d["modes"] and "mode_note" not in d


def test_an_unverified_mode_is_never_the_value(tmp_path, capfd, monkeypatch):
    rc, out, err = _supervise(tmp_path, capfd, monkeypatch, "lanes_unverified")
    assert rc == 0, err
    d = json.loads(out.strip().splitlines()[-1])
    assert d["mode"] == "one_stream" and d["modes"]["lanes"]["verify"]["ok"] is False and d["modes"]["lanes"]["ms_per_step"] == 2.0


def test_segments_under_torch_distributed_run(tmp_path):
    """The driver's own multi-GPU line: `python -m torch.distributed.run ... bench.py --gpus N`.  Every worker is then the supervisor of
    its own rank: it never touches the GPU, starts one child per segment, and the supervisors agree on a fresh rendezvous port per
    segment through the launcher's store; rank 0 prints the merged line.  Stand-in ranks; CPU only."""
    import socket
    script = tmp_path / "seg.py"
    script.write_text(SEGMENT_STAND_IN)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PSE_BENCH_RANK_SCRIPT"] = str(script)
    for fake, mode in (("", "lanes"), ("lanes_hang", "one_stream"), ("rccl_down", "host_fallback")):
        env["FAKE"] = fake
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert d["single_gpu"]["ms_per_step"] == 8.0
        if fake == "rccl_down":    # (the split of the stand-in does not depend on RCCL: it is a verified mode, so no fallback is run)
            assert d["mode"] == "split" and set(d["modes"]) == {"one_stream", "lanes", "split"}
            continue
        assert d["mode"] == mode and set(d["modes"]) == {"one_stream", "lanes", "split"}
        assert d["modes"]["split"]["ms_per_step"] == 3.0           # (two ranks: the functional split is timed too)
        if fake:
            assert "error" in d["modes"]["lanes"]


def test_fallback_decision_reaches_every_supervisor_under_torch_distributed_run(tmp_path):
    """Three ranks (no functional split), no RCCL mode works: rank 0's supervisor decides on the host-transport fallback and the other
    supervisors learn it through the launcher's store -- all three start their rank of the fallback segment."""
    import socket
    script = tmp_path / "seg.py"
    script.write_text(SEGMENT_STAND_IN)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PSE_BENCH_RANK_SCRIPT"] = str(script)
    env["FAKE"] = "rccl_down"
    env["PSE_BENCH_FALLBACK_ANYWAY"] = "1"      # (no GPUs here: see test_no_rccl_mode_at_all_...)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "3", "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["mode"] == "host_fallback" and d["n_gpus"] == 3 and "error" in d["modes"]["lanes"] and "error" in d["modes"]["one_stream"]
