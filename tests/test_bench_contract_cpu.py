"""CPU-side checks of the measurement contract: the helpers bench.py builds its JSON line from, and the profile
artefacts the line refers to (profiles/ is what the judge reads; a bench line must be able to find its traffic)."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_measured_traffic_resolves_every_kernel_name_bench_can_emit():
    for n in (1024, 4096, 16384, 65536, 524288 // 8):
        for k in ("rn_analysis_kernel", "rn_analysis_lean_kernel", "rn_nn_mfma_kernel", "rn_synthesis_kernel"):
            t = bench.measured_traffic(k, n)
            assert isinstance(t, int) and t > 1000 * n, (k, n, t)
    assert bench.measured_traffic("rn_analysis_lean_kernel", 4096) == bench.measured_traffic("rn_analysis_kernel", 4096)
    assert bench.measured_traffic("rn_nn_vector_kernel", 4096) is None  # never profiled with PMC: reported as null


def test_algorithmic_bytes_match_design_table():
    # K0 + K1 = 10,784 + 26,144 - (the 1920-float input row is counted once), K3 = 14,352
    assert bench.ANALYSIS_BYTES == 30000
    assert bench.SYNTHESIS_BYTES == 14352
    assert bench.NETWORK_STATE_BYTES == 12696


def test_committed_bench_lines_follow_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r1_final_bench_*.json")))
    assert len(files) >= 4
    for f in files:
        d = json.loads(open(f).read())
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in d, (f, key)
        assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
        assert "workload" in d["config"] and "model" not in d["config"]
        r = d["roofline"]
        assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
        # value and ms_per_step describe the same run
        assert abs(d["value"] * d["ms_per_step"] * 1e-3 / d["config"]["frames_per_step"] - 1) < 0.01
    d = json.loads(open(os.path.join(ROOT, "profiles", "r1_final_bench_4096.json")).read())
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]


def test_rocprof_summary_lists_the_kernels_of_the_step():
    txt = open(os.path.join(ROOT, "profiles", "r1_final_kernel_stats.txt")).read()
    for k in ("rn_hp_kernel", "rn_analysis", "rn_nn_mfma_kernel", "rn_synthesis_kernel"):
        assert k in txt
