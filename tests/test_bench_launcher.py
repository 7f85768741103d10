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
    # the layer's exchange step (row-parallel o_proj, chunked all-reduce) ran over gloo and equalled the ranks' sum
    x = j["extra"]
    assert x["oproj_gemm_allreduce_ms"] > 0 and "layer_ms_with_oproj" in x and x["oproj"]["chunk_tokens"] == 64


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
    # a rank that dies takes the job down with a non-zero exit code, and the launcher shows that rank's stderr
    r = _bench("--gpus", "2", "--dry-run", "--model", "8b", "--seq-len", "-5")
    assert r.returncode != 0 and "Its stderr ends" in r.stderr and "Traceback" in r.stderr


def test_the_launcher_counts_gpus_without_touching_one():
    """The parent process must stay GPU-free (it goes on to start one process per GPU): the device count comes
    from the visibility variables or from the kernel driver's topology files, never from a HIP call."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    keep = {k: os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")}
    try:
        for k in keep:
            os.environ.pop(k, None)
        os.environ["HIP_VISIBLE_DEVICES"] = "0,1,2"
        assert bench.visible_gpu_count() == 3
        os.environ["CUDA_VISIBLE_DEVICES"] = "1"
        assert bench.visible_gpu_count() == 1
        os.environ["HIP_VISIBLE_DEVICES"] = ""
        assert bench.visible_gpu_count() == 0
        for k in keep:
            os.environ.pop(k, None)
        n = bench.visible_gpu_count()  # topology files (or None where the driver is absent)
        assert n is None or n >= 0
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def launch_ranks"):src.index("def dry_run")]
    assert "torch.cuda" not in body and "device_count" not in body
