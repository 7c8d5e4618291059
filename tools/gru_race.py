#!/usr/bin/env python3
"""Race hunt for the GRU layer kernel (nn_layers.hip): which variant / schedule / wait discipline loses bits, how often, and why.

Every configuration runs in its own process (the switches are environment variables the library reads once):
N streams = replicas of a 32-stream block through pipelined calls of 5 + 1 + 8 frames, `reps` times on ONE batch (reset in
between); after every repetition the replicas are compared on the GPU (gains, vad, pcm).  The checking instantiations of the
instrumented build (w4chk, w8b1chk, w8chk) additionally compare every h_old vector a lane takes from its LDS row buffer with
the same bytes loaded straight from HBM and say whether a wrong one was the previous unit tile's (a stale buffer).

usage: tools/gru_race.py [--streams 32768] [--model little] [--reps 100] [--only name,name] > profiles/r5_gru_race.txt
       tools/gru_race.py --worker ...   (internal)"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

# name -> (environment, instrumented build?)
# "round-4 K0" = the high-pass kernel as it shipped until round 4, built WITH the SLP vectoriser (hp_slp.hip: packed multiplies with
# op_sel operand selects); it only exists in the instrumented library ($RNNOISE_AMD_HP_AB=2048).  The product's K0 has no packed math.
OLD_K0 = {"RNNOISE_AMD_HP_AB": "2048"}
W4, W8, HPSIDE = {"RNNOISE_AMD_GRU_VARIANT": "w4"}, {"RNNOISE_AMD_GRU_VARIANT": "w8"}, {"RNNOISE_AMD_PIPE": "1"}
CONFIGS = {
    # the product library
    "product: w8 layer kernel (8 waves, 152 KB: what shipped until round 4), three-stream pipeline": (W8, False),
    "product: w8, only the high-pass on a side stream (the host-fed path's schedule)": ({**W8, **HPSIDE}, False),
    "product: w4 layer kernel (4 waves, 72 KB: shares SIMDs with other kernels' waves; the default since round 5), three-stream pipeline": (W4, False),
    "product: w4, only the high-pass on a side stream": ({**W4, **HPSIDE}, False),
    "product: w4b2 (4 waves, two row buffers), only the high-pass on a side stream": ({"RNNOISE_AMD_GRU_VARIANT": "w4b2", **HPSIDE}, False),
    # the failure of round 4, reproduced: the same library with the round-4 K0
    "round-4 K0 + w4, three-stream pipeline (round 4's unstable configuration)": ({**W4, **OLD_K0}, True),
    "round-4 K0 + w4, only the high-pass on a side stream": ({**W4, **HPSIDE, **OLD_K0}, True),
    "round-4 K0 + w4, everything on one stream (K0 never beside the layer kernel)": ({**W4, "RNNOISE_AMD_PIPE": "9", **OLD_K0}, True),
    "round-4 K0 + w8, only the high-pass on a side stream (a w8 workgroup leaves no registers for a K0 wave on its SIMDs)": ({**W8, **HPSIDE, **OLD_K0}, True),
    # what it is not
    "round-4 K0 + w4 hp-side, a workgroup barrier between the layer kernel's vmcnt(0) and its row reads": ({**W4, **HPSIDE, **OLD_K0, "RNNOISE_AMD_GRU_SETTLE": "2"}, True),
    "round-4 K0 + w4 hp-side, ordering events WITH the system-scope fence": ({**W4, **HPSIDE, **OLD_K0, "RNNOISE_AMD_EVENT_FENCE": "1"}, True),
    "round-4 K0 + w4nodma hp-side (the layer kernel without LDS-DMA: pieces through registers and ds_write)": ({"RNNOISE_AMD_GRU_VARIANT": "w4nodma", **HPSIDE, **OLD_K0}, True),
    "round-4 K0 + w4 hp-side, the layer kernel's activations element by element (no packed math in IT)": ({**W4, **HPSIDE, **OLD_K0, "RNNOISE_AMD_GRU_ACT": "0"}, True),
    "round-4 K0 + w4 hp-side, layer kernel without s_setprio": ({**W4, **HPSIDE, **OLD_K0, "RNNOISE_AMD_GRU_PRIO": "0"}, True),
    "round-4 K0 + w4 hp-side, K0 drains its stores before it reads the pitch ring back": ({**W4, **HPSIDE, "RNNOISE_AMD_HP_AB": str(2048 + 256)}, True),
    "round-4 K0 + w4 hp-side, K0 drains its tap stores before it ends": ({**W4, **HPSIDE, "RNNOISE_AMD_HP_AB": str(2048 + 1024)}, True),
    # the layer kernel's own inputs, checked word by word against HBM while the failure happens
    "round-4 K0 + w4chk hp-side (every LDS row / image word the layer kernel uses compared with HBM)": ({"RNNOISE_AMD_GRU_VARIANT": "w4chk", **HPSIDE, **OLD_K0}, True),
    "w8chk, three-stream pipeline (the eight-wave kernel's rows and images compared with HBM)": ({"RNNOISE_AMD_GRU_VARIANT": "w8chk"}, True),
}
FIELDS = (("analysis_mem", 0, 480), ("synthesis_mem", 480, 480), ("pitch_buf", 960, 1728), ("last_gain", 2688, 1), ("last_period", 2689, 1),
          ("mem_hp", 2690, 2), ("lastg", 2692, 32), ("conv1_state (the last two frames' features)", 2724, 130), ("conv2_state", 2854, 256),
          ("gru1", 3110, 384), ("gru2", 3494, 384), ("gru3", 3878, 384), ("delayed X", 4262, 962), ("delayed P", 5224, 962),
          ("delayed Ex/Ep/Exp", 6186, 96))


def worker(a):
    import contextlib
    import ctypes as C
    import numpy as np, torch
    from rnnoise_amd import capi, synth
    from conftest import load_blob
    N, reps = a.streams, a.reps
    ctx = capi.instrumented() if a.instr else contextlib.nullcontext()
    with ctx as L:
        capi.set_rcp_profile("intel")
        calls = (5, 1, 8); T = sum(calls)
        base = synth.batch_pcm(range(32), T); base[:3, 9] = 0; base[T - 5:T - 3, 12] = 0
        dev = torch.device("cuda", 0)
        d_in = torch.from_numpy(base).to(dev).repeat(1, N // 32, 1).contiguous()
        d_out = torch.empty_like(d_in); d_vad = torch.empty((T, N), device=dev); d_gains = torch.empty((T, N, 32), device=dev)
        m = capi.Model(load_blob(a.model)); b = capi.Batch(m, N); b.set_nn_path(1)
        st = torch.cuda.current_stream().cuda_stream
        if a.instr:
            log = (C.c_uint * 484)(); L.rnnoise_amd_debug_gru_race(0, log, 484)  # clear
        bad_reps, notes, first = 0, [], None
        def run():
            b.reset(); f = 0
            for n in calls:
                b.process_device(d_out[f].data_ptr(), d_in[f].data_ptr(), d_vad[f].data_ptr(), d_gains[f].data_ptr(), n, st); f += n
            torch.cuda.synchronize()
        # the reference every repetition is compared with: the same calls with every kernel on ONE stream
        old = b.set_schedule(9); run(); b.set_schedule(old)
        first = [t.clone() for t in (d_gains, d_vad, d_out)]
        assert all(bool((t.view(torch.int32).reshape(T, N // 32, -1) == t.view(torch.int32).reshape(T, N // 32, -1)[:, :1]).all()) for t in first), "one-stream run diverged"
        t0 = time.time()
        for rep in range(reps):
            run()
            msg = []
            for name, t, w, ref in (("gains", d_gains, 32, first[0]), ("vad", d_vad, 1, first[1]), ("pcm", d_out, 480, first[2])):
                r = t.view(torch.int32).reshape(T, N // 32, 32, w)
                bad = (r != r[:, :1]).any(dim=3)
                if bool(bad.any()):
                    idx = bad.nonzero()
                    fr = sorted(set(idx[:, 0].tolist())); rp = sorted(set(idx[:, 1].tolist())); ss = sorted(set(idx[:, 2].tolist()))
                    msg.append(f"{name}: frames {fr[0]}..{fr[-1]} replicas {rp[:6]} (n={len(rp)}) streams {ss[:12]}")
                    if name == "gains" and len(notes) < 2:  # which part of the state of one wrong stream differs from its twin in replica 0
                        fr0, r0, s0 = idx[0].tolist()
                        twin = 0 if r0 else 1
                        sa, sb = b.export_state(32 * r0 + s0).view(np.uint32), b.export_state(32 * twin + s0).view(np.uint32)
                        parts = []
                        for fname, off, ln in FIELDS:
                            d = np.nonzero(sa[off:off + ln] != sb[off:off + ln])[0]
                            if len(d): parts.append(f"{fname}: {len(d)} of {ln} words (first {d[0]}, last {d[-1]})")
                        msg.append(f"state of stream {32 * r0 + s0} vs {32 * twin + s0} after the run -> " + ("; ".join(parts) or "identical"))
                elif not torch.equal(t.view(torch.int32), ref.view(torch.int32)):
                    msg.append(f"{name}: replicas agree but differ from the one-stream run")
            if msg:
                bad_reps += 1
                if len(notes) < 3: notes.append(f"rep {rep}: " + "; ".join(msg))
        res = {"reps": reps, "bad_reps": bad_reps, "s": round(time.time() - t0, 2), "notes": notes}
        if a.instr:
            L.rnnoise_amd_debug_gru_race(0, log, 484)
            w = list(log)
            res["row_check"] = {"vectors_checked_mod_2^32": w[2], "differ_from_hbm": w[0] - w[3], "of_them_stale_previous_tile": w[1], "image_words_differ": w[3]}
            recs = []
            for k in range(min(w[0], 40))[:10]:
                r = w[4 + 12 * k: 16 + 12 * k]
                if r[1] >> 16 == 0xffff:
                    recs.append({"block": r[0], "when": "end" if r[1] & 0x100 else "start", "region": ("xq%d" % (r[1] & 3), "hq%d" % (r[1] & 3), "table")[(r[1] & 255) // 4],
                                 "word": r[2], "got": hex(r[3]), "hbm": hex(r[4]), "wave": r[5]})
                else:
                    recs.append({"block": r[0], "wave": r[1] & 255, "ui": (r[1] >> 8) & 255, "tile": (r[1] >> 16) & 255, "lane": r[1] >> 24,
                                 "got": [hex(x) for x in r[2:6]], "hbm": [hex(x) for x in r[6:10]], "prev_tile": [hex(x) for x in r[10:12]]})
            res["row_check"]["records"] = recs
        b.close(); m.close()
    print("RESULT " + json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=32768); ap.add_argument("--model", default="little")
    ap.add_argument("--reps", type=int, default=100); ap.add_argument("--only", default="")
    ap.add_argument("--worker", action="store_true"); ap.add_argument("--instr", type=int, default=0)
    a = ap.parse_args()
    if a.worker:
        return worker(a)
    print(f"# tools/gru_race.py: {a.streams} streams ({a.model} model) = replicas of a 32-stream block, calls of 5 + 1 + 8 frames, {a.reps} repetitions per "
          f"configuration on one batch;\n# bad = repetitions in which replicas diverged (or differed from repetition 0)")
    for name, (env, instr) in CONFIGS.items():
        if a.only and not any(k in name for k in a.only.split(",")):
            continue
        e = dict(os.environ); e.update(env)
        # (the switches under test -- forms of the layer kernel, $RNNOISE_AMD_HP_AB, _GRU_SETTLE ... -- exist in the instrumented library only)
        e.setdefault("RNNOISE_AMD_LIB", os.path.join(ROOT, "rnnoise_amd", "librnnoise_amd_instr.so"))
        r = subprocess.run([sys.executable, __file__, "--worker", "--streams", str(a.streams), "--model", a.model, "--reps", str(a.reps),
                            "--instr", str(int(instr))], env=e, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(f"{name}: FAILED rc {r.returncode}: {r.stderr.strip()[-300:]}", flush=True)
            continue
        res = json.loads(line[0][7:])
        print(f"{name}\n    env {env or '{}'}: bad {res['bad_reps']} / {res['reps']}  ({res['s']} s)", flush=True)
        for n in res["notes"]:
            print("      " + n)
        if "row_check" in res:
            rc = res["row_check"]
            print(f"      row check: {rc['differ_from_hbm']} vectors differed from HBM, {rc['of_them_stale_previous_tile']} of them = the previous unit "
                  f"tile's vector (stale buffer); {rc['vectors_checked_mod_2^32']} checked (mod 2^32); image words (input, state, table) that differed from HBM: {rc['image_words_differ']}")
            for rec in rc["records"]:
                print(f"        {rec}")


if __name__ == "__main__":
    main()
