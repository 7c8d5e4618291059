"""CPU-side checks of the measurement contract: the helpers bench.py builds its JSON line from, and the profile
artefacts the line refers to (profiles/ is what the judge reads; a bench line must be able to find its traffic)."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_pmc_record_resolves_every_kernel_name_bench_can_emit():
    for n in (1024, 4096, 8192, 10240, 16384, 65536, 524288 // 8):
        names = [bench.kernel_of(kind, n) for kind in ("highpass", "analysis", "network", "synthesis")]
        # five launches from 10,240 streams up; the four-wave layer kernel once there are more 64-stream groups than CUs
        assert names[2] == ("rn_nn_gru_kernel" if n > 16384 else ("rn_nn_gru_w8_kernel" if n >= 10240 else "rn_nn_mfma_kernel"))
        assert names[0] == ("rn_hp_one_kernel" if n <= 2048 else "rn_hp_kernel")  # one wave per stream up to 2048 streams
        for k in names + ["rn_analysis_kernel", "rn_analysis_single_kernel"]:
            if k == "rn_hp_one_kernel":
                continue  # (no PMC pass at a batch size that runs it: its traffic is reported as null)
            r = bench.pmc_record(k, n)
            assert r and r["hbm_bytes_per_frame"] > 1000 and r["valu_per_wave"] > 100, (k, n, r)
    for k in bench.NN_LAYER_KERNELS:  # every launch of the layer-wise network has its own PMC record
        assert bench.pmc_record(k, 65536)["hbm_bytes_per_frame"] > 1000, k
    assert bench.pmc_record("rn_nn_vector_kernel", 4096) is None  # never profiled with PMC: traffic is reported as null
    assert bench.kernel_of("synthesis", 256) == "rn_synthesis_few_kernel" and bench.kernel_of("synthesis", 257) == "rn_synthesis_kernel"


def test_algorithmic_bytes_match_design_table():
    # DESIGN.md section 4
    assert bench.ALG_BYTES == {"highpass": 7344, "analysis": 22716, "network": 12696, "synthesis": 14352}
    assert bench.waves_per_launch("analysis", 65536) == 65536 and bench.waves_per_launch("highpass", 65536) == 1024
    assert bench.waves_per_launch("network", 65536) == 1024 * 8 and bench.waves_per_launch("network", 17) == 16


def test_step_valu_issue_bound_from_the_pmc_tables():
    """bench.py's roofline_step_valu: the step's kernels x their measured VALU instruction counts x the mix-priced issue cost
    of each kernel (3.0 - 3.7 clk: profiles/valu_mix.json over profiles/r3_valu_issue.txt); 65,536 streams on the layer-wise
    network come to about 1.1 ms of pure issue time, the vector path has no PMC record"""
    v = bench.step_valu_issue_ms(65536)
    assert 0.9 < v < 1.5, v
    assert abs(bench.step_valu_issue_ms(32768, "little") / v - 0.5) < 0.05
    assert 0.04 < bench.step_valu_issue_ms(4096) < 0.2
    assert 2.26 <= bench.valu_cost("rn_analysis_kernel") <= 4.15 and bench.valu_cost("no_such_kernel") == 4.15
    assert bench.step_valu_issue_ms(4096, "default", "vector") is None


def test_defaults_are_the_largest_single_gpu_config():
    a = bench.parse_args([])
    assert a.streams == 65536 and a.gpus == 1 and a.repeats >= 25 and a.nn == "mfma" and a.model == "default"
    assert "configs[2]" in bench.workload_name(a, a.streams)
    a = bench.parse_args(["--model", "little", "--streams", "32768"])
    assert "configs[3]" in bench.workload_name(a, a.streams)
    a = bench.parse_args(["--gpus", "8"])
    assert "configs[4]" in bench.workload_name(a, a.streams)


def test_usable_cpus_respects_affinity_and_quota(monkeypatch):
    n = bench.usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    monkeypatch.setattr(bench, "cgroup_cpu_quota", lambda: 3)
    assert bench.usable_cpus() == min(3, len(os.sched_getaffinity(0)))


def _check_line(d, f):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, (f, key)
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # value and ms_per_step describe the same run
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / d["config"]["frames_per_step"] - 1) < 0.01


def test_committed_bench_lines_follow_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r1_final_bench_*.json")))
    assert len(files) >= 4
    for f in files:
        _check_line(json.loads(open(f).read()), f)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r2_*bench_*.json"))):
        d = json.loads(open(f).read())
        _check_line(d, f)
        assert d["repeats"] >= 5 and d["value_min"] <= d["value"] <= d["value_max"], f
        # numerator and denominator of the roofline cover the same kernel (round-1 ADVICE): its own bytes / its own time
        r = d["roofline"]
        kind = {"rn_hp_kernel": "highpass", "rn_analysis_kernel": "analysis", "rn_analysis_single_kernel": "analysis",
                "rn_nn_mfma_kernel": "network", "rn_nn_gru_kernel": "network", "rn_nn_vector_kernel": "network",
                "rn_synthesis_kernel": "synthesis"}[r["kernel"]]
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"][kind] * 1e-3) / 1e9) < 0.02 * r["achieved"]
        if d["n_gpus"] == 1 and "cpu_baseline" in d:
            cb = d["cpu_baseline"]
            assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
            assert cb["one_thread_frames_per_s"] > 0 and cb["frames_per_s_per_core"] > 0


def test_rocprof_summary_lists_the_kernels_of_the_step():
    for f in ["r1_final_kernel_stats.txt"] + [os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "profiles", "r2_*kernel_stats*.txt"))]:
        txt = open(os.path.join(ROOT, "profiles", f)).read()
        nn = ("rn_nn_mfma_kernel",) if f.startswith("r1_") else ("rn_nn_front_kernel", "rn_nn_gru_kernel", "rn_nn_dense_kernel")
        for k in ("rn_hp_kernel", "rn_analysis", "rn_synthesis_kernel") + nn:
            assert k in txt, (f, k)


def test_new_bench_flags_and_valu_mix_table():
    """round 3: --s16 / --frames-per-call / --no-parity parse; the per-kernel VALU price table (tools/valu_mix.py ->
    profiles/valu_mix.json) covers every kernel of the step and stays between the measured class costs"""
    a = bench.parse_args(["--s16", "--frames-per-call", "1", "--host-io", "--no-parity"])
    assert a.s16 and a.frames_per_call == 1 and a.host_io and a.no_parity
    assert "int16" in bench.workload_name(a, a.streams) and "1 frame(s) per call" in bench.workload_name(a, a.streams)
    mix = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
    lo, hi = mix["cost"]["fast"], mix["cost"]["trans"]
    for k in ("rn_hp_kernel", "rn_hp_one_kernel", "rn_analysis_kernel", "rn_analysis_single_kernel", "rn_nn_front_kernel",
              "rn_nn_gru_kernel", "rn_nn_dense_kernel", "rn_nn_mfma_kernel", "rn_nn_one_kernel", "rn_nn_vector_kernel",
              "rn_synthesis_kernel"):
        r = mix["kernels"][k]
        assert lo <= r["mean_cycles"] <= hi and r["fast"] + r["std"] + r["trans"] == r["valu_static"], k
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import valu_mix
    assert valu_mix.classify("v_fma_f32", "v1, v2, v3, v4") == "fast"
    assert valu_mix.classify("v_add_f32_dpp", "v1, v2, v3 row_shr:1") == "std"
    assert valu_mix.classify("v_add_f32_e32", "v1, s4, v3") == "std"       # an SGPR source costs the slow rate
    assert valu_mix.classify("v_rcp_f32_e32", "v1, v2") == "trans" and valu_mix.classify("v_permlane32_swap_b32_e32", "v1, v2") == "trans"
    assert valu_mix.classify("v_cndmask_b32_e32", "v1, v2, v3, vcc") == "std"


def test_round3_bench_lines_carry_parity_and_the_new_rooflines():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r3_bench_*.json")))
    assert len(files) >= 8
    for f in files:
        d = json.loads(open(f).read())
        _check_line(d, f)
        assert d["parity"]["bit_identical"] is True and d["parity"]["streams"] == 32, f
        if "roofline_lds" in d:
            assert 0 < d["roofline_lds"]["frac"] < 1 and 0 < d["roofline_valu"]["frac_f32_peak"] <= d["roofline_valu"]["frac"] < 1, f


def test_round4_rooflines_are_measurements():
    """round 4: the VALU / LDS rooflines of the bench line stand on measured inputs -- the K1 instruction mix is weighted by the
    per-section instruction counts of the PMC passes (not by the static listing), the shader clock comes from GRBM_GUI_ACTIVE
    over the kernel's duration (one figure, <= the 2.4 GHz the part can reach), and the committed r4 lines say which they used"""
    mix = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
    k1 = mix["kernels"]["rn_analysis_kernel"]
    assert k1["weighting"].startswith("dynamic") and len(k1["sections"]) == 17
    assert sum(s["valu"] for s in k1["sections"]) == k1["valu_dynamic_per_wave"]
    assert mix["cost"]["fast"] <= k1["mean_cycles"] <= mix["cost"]["trans"]
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_by_streams.json")))
    for n, rec in pmc["by_streams"].items():
        for name, r in rec.items():
            if isinstance(r, dict) and "clock_ghz" in r:
                assert 1.5 <= r["clock_ghz"] <= 2.4 or name != "rn_analysis_kernel", (n, name, r["clock_ghz"])
    assert "clock_ghz" in pmc["by_streams"]["65536"]["rn_analysis_kernel"]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r4_bench_*.json")))
    assert len(files) >= 8
    for f in files:
        d = json.loads(open(f).read())
        _check_line(d, f)
        assert d["parity"]["bit_identical"] is True, f
    d = json.loads(open(os.path.join(ROOT, "profiles", "r4_bench_65536.json")).read())
    v = d["roofline_valu"]
    assert v["clocks_per_inst_kind"] == "dynamic" and "GRBM_GUI_ACTIVE" in v["clock_source"] and 1.5 <= v["clock_ghz"] <= 2.4
    assert d["roofline_lds"]["clock_ghz"] == v["clock_ghz"]
    assert d["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline"]["cores"] >= 1
