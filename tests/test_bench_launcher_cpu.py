"""CPU suite: bench.py's OWN multi-rank code path -- self-launch under torch.distributed.run, RANK / WORLD_SIZE
handling, stream sharding, the max-over-ranks reduction and the rank-0-only JSON line -- exercised with gloo and a stub
batch (`--stub`: no GPU exists here; the stub computes nothing and its line says so).  The product path of the same
function is what runs on the GPU box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def test_direct_invocation_with_gpus_2_self_launches_and_prints_one_line():
    r = _run(["--gpus", "2", "--stub", "--streams", "96", "--steps", "3", "--warmup", "1", "--repeats", "5"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    d = json.loads(lines[0])
    assert d["stub"] is True and "SELF-TEST" in d["metric"]
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["repeats"] == 5
    assert d["config"]["streams_per_gpu"] == 96 and d["config"]["frames_per_step"] == 192
    assert d["config"]["stream_ids_rank0"] == [0, 96]       # rnnoise_amd.dist.shard_streams
    assert d["scaling"] == "weak" and d["value_min"] <= d["value"] <= d["value_max"]
    # whole-job aggregate: frames of BOTH ranks over the max-over-ranks time of the median repetition
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / 192 - 1) < 0.01
    assert "cpu_baseline" not in d                            # rank 0 at N=1 only


def test_single_rank_stub_line():
    r = _run(["--stub", "--streams", "64", "--steps", "2", "--warmup", "1", "--repeats", "5", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["frames_per_step"] == 64


def test_without_a_gpu_the_real_path_refuses_loudly():
    r = _run(["--streams", "64", "--steps", "1", "--warmup", "0", "--repeats", "1", "--no-cpu-baseline"])
    import torch
    if torch.cuda.is_available():
        return
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)


def test_a_rank_without_a_working_data_communicator_sends_the_job_to_gloo():
    """the 8-GPU run's first collective must not be able to kill or hang it: every rank tries the data group once and votes
    over the gloo control group; one failing rank (injected here) -> "collective": "gloo-fallback", the line still comes out"""
    args = ["--gpus", "2", "--stub", "--streams", "64", "--smoke"]
    ok = _run(args, {"RNNOISE_AMD_BENCH_DATA_BACKEND": "gloo"})
    assert ok.returncode == 0, ok.stdout[-2000:] + ok.stderr[-3000:]
    d = json.loads([l for l in ok.stdout.splitlines() if l.startswith("{")][-1])
    assert d["collective"] == "gloo" and "collective_error" not in d
    assert d["steps"] == 2 and d["warmup"] == 1 and d["repeats"] == 1          # --smoke
    assert len(d["value_by_rank"]) == 2 and all(v > 0 for v in d["value_by_rank"])
    assert d["value"] <= sum(d["value_by_rank"]) * 1.001                        # the aggregate is paced by the slowest rank
    bad = _run(args, {"RNNOISE_AMD_BENCH_DATA_BACKEND": "gloo", "RNNOISE_AMD_BENCH_BREAK_RCCL": "1", "RNNOISE_AMD_BENCH_DATA_TIMEOUT": "8"})
    assert bad.returncode == 0, bad.stdout[-2000:] + bad.stderr[-3000:]
    d = json.loads([l for l in bad.stdout.splitlines() if l.startswith("{")][-1])
    # (the error quoted is the first in rank order: rank 0's time-out waiting for rank 1, whose communicator "failed")
    assert d["collective"] == "gloo-fallback" and d["collective_error"].startswith("rank ") and d["n_gpus"] == 2
