"""bench.py's own launcher (the driver calls `python bench.py --gpus N`, no torchrun): the parent must start N
fresh rank processes with the rendezvous environment set and print exactly ONE JSON line.  Exercised here with
--dry-launch (gloo on CPU, no filter: there is no CPU filter to benchmark)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    return env


def test_self_launch_two_ranks_one_json_line():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["dry_launch"] and j["launched_by"] == "bench.py launcher"
    envs = j["rank_env"]
    assert [e["RANK"] for e in envs] == ["0", "1"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1"]
    assert all(e["WORLD_SIZE"] == "2" and e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    # set in the children's environment before they start, i.e. before any HIP initialisation
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)


def test_launcher_parent_never_imports_torch():
    """The launcher process must not initialise the GPU: it may not even import torch."""
    code = ("import sys, runpy; sys.argv=['bench.py','--gpus','2','--dry-launch','--steps','1'];\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    rc = e.code\n"
            "assert 'torch' not in sys.modules, 'launcher imported torch'; sys.exit(rc or 0)" % BENCH)
    p = subprocess.run([sys.executable, "-c", code], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len([l for l in p.stdout.splitlines() if l.strip()]) == 1


def test_failed_rank_fails_the_launch():
    env = _clean_env()
    env["PHL_BENCH_TEST_FAIL_RANK"] = "1"
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--dry-launch"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0


def test_torchrun_form_still_works():
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29647", BENCH, "--gpus", "2", "--steps", "2",
                        "--dry-launch"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["launched_by"] != "bench.py launcher"
