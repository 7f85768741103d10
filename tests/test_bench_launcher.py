"""bench.py --gpus N starts its own ranks (no GPU needed for the dry run: gloo rendezvous, host-side sizing)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], cwd=ROOT, env=e, capture_output=True,
                          text=True, timeout=300)


def test_gpus_n_launches_n_ranks_by_itself():
    r = _bench("--gpus", "2", "--dry-run")
    assert r.returncode == 0, r.stderr
    (line,) = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["dry_run"] is True and j["scaling"] == "weak" and j["value"] is None
    c = j["config"]
    assert (c["global_batch"], c["heads_per_gpu"], c["kv_heads_per_gpu"], c["seq_len"]) == (2, 16, 4, 32768)
    assert c["parallelism"] == "tp2 (heads)" and "Qwen3-8B" in j["metric"] and "32k" in j["metric"]


def test_model_configurations_of_the_tp_rows():
    r = _bench("--gpus", "4", "--model", "32b", "--dry-run")
    assert r.returncode == 0, r.stderr
    c = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert c["n_gpus"] == 4 and (c["config"]["heads_per_gpu"], c["config"]["kv_heads_per_gpu"]) == (16, 2)
    assert "Qwen3-32B" in c["config"]["workload"] and c["config"]["seq_len"] == 32768
    # one rank of the fp8 row alone (its shard is sized as rank 0 of an 8-rank job is)
    r = _bench("--gpus", "1", "--model", "72b-fp8", "--dry-run")
    assert r.returncode == 0, r.stderr
    c = json.loads(r.stdout.splitlines()[-1])
    assert c["n_gpus"] == 1 and c["config"]["seq_len"] == 65536 and "fp8" in c["config"]["workload"]


def test_too_few_devices_or_ranks_is_an_error_not_a_single_gpu_number():
    r = _bench("--gpus", "2", env={"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert r.returncode != 0 and "HIP device" in r.stderr and not r.stdout.strip()
    # a launcher that started fewer ranks than --gpus says: refuse instead of reporting n_gpus 1
    r = _bench("--gpus", "4", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "--gpus 4" in r.stderr
    # a rank that dies takes the job down with a non-zero exit code
    r = _bench("--gpus", "2", "--dry-run", "--model", "8b", "--seq-len", "-5")
    assert r.returncode != 0
