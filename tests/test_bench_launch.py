"""`python bench.py --gpus N` as the driver types it: with no launcher in the environment the script
starts torch.distributed.run itself as a child process and relays ONE JSON line.  Exercised on CPU with the
`stub` workload (gloo, no HIP kernel) so the launch, the barrier / max-over-ranks timing and the all-gather of
bench.py itself run at world size 2 without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def test_gpus_2_launches_itself():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "stub"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert len(out["per_rank_pairs_per_s"]) == 2 and all(v > 0 for v in out["per_rank_pairs_per_s"])
    assert out["value"] > 0 and out["config"]["workload"] == "stub"


def test_single_rank_stub_and_child_failure_code():
    r = _run(["--steps", "2", "--warmup", "0", "--workload", "stub"])
    assert r.returncode == 0 and json.loads(r.stdout)["n_gpus"] == 1
    # a failing rank's exit code comes back through the launcher (no GPU here: the real workload refuses to start)
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--only", "--no-cpu-baseline"],
             env_extra={"HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    assert r.returncode != 0 and not r.stdout.strip()


def test_rank_count_mismatch_is_refused():
    r = _run(["--gpus", "4", "--workload", "stub"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
