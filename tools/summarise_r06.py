"""Turn gpurun_out/<tag>/ (written on the GPU box by tools/evidence_r06.sh) into the tracked files profiles/<tag>_*.   usage: python tools/summarise_r06.py r06
Per kernel CLASS of rsb_step_kernel (the resident class <.., 64, ..> = `value`; the pipelined twin <.., 16, ..>; the plain class <.., 0, ..> = lock-step):
launch durations from rocprofv3 --kernel-trace next to the bench line's HIP-event brackets, HBM bytes from the FETCH_SIZE / WRITE_SIZE passes (calibrated
2 x FETCH + WRITE, profiles/r02_traffic_calibration.txt) - per launch for the per-control-step classes, per CONTROL STEP for the resident class -, SQ counters."""
import collections, csv, glob, json, os, re, shutil, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
O, P = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")


def F(label, ext=""):
    hits = sorted(glob.glob(os.path.join(O, f"[0-9][0-9]_{label}{ext}")))
    if not hits:
        raise SystemExit(f"no output of a step labelled {label!r} under {O}")
    return hits[-1]


cp = lambda src, dst: shutil.copy(src, os.path.join(P, f"{tag}_{dst}"))
cp(F("default20", ".json"), "bench_default_line_steps20_warmup5.json")
cp(F("default300", ".json"), "bench_default_line.json")
cp(F("c3", ".json"), "bench_config3.json"); cp(F("c5", ".json"), "bench_config5.json")
cp(F("closed20", ".json"), "bench_closed_loop_steps20_warmup5.json"); cp(F("closed300", ".json"), "bench_closed_loop.json")
for c in (2, 3, 5):
    cp(os.path.join(F(f"trace_c{c}"), "run_kernel_stats.csv"), f"rocprofv3_kernel_stats_config{c}.csv")
cp(os.path.join(F("trace_closed"), "run_kernel_stats.csv"), "rocprofv3_kernel_stats_closed_loop.csv")
open(os.path.join(P, f"{tag}_pytest_gpu_tail.txt"), "w").write("".join(open(F("pytest", ".log")).readlines()[-16:]) + open(F("smoke", ".log")).read())
shutil.copy(F("prof_template", ".log"), os.path.join(P, f"{tag}_template_path_host_evidence_run.txt"))


def klass(name):
    m = re.search(r"rsb_step_kernel<(\d+), (\d+), (\d+), (\d+)", name)
    if not m:
        return None
    cl = int(m.group(3))
    return ("resident" if cl & 64 else "pipelined twin" if cl & 16 else "plain") + f" <{m.group(1)},{m.group(2)},{cl},{m.group(4)}>"


def counters(sub):
    fs = glob.glob(os.path.join(F(sub), "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = klass(r["Kernel_Name"])
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


ABYTES = {2: 456.0, 3: 520.0, 5: 1080.0}
lines = []
for c in (2, 3, 5):
    b = json.load(open(F("default20", ".json"))) if c == 2 else json.load(open(F(f"c{c}", ".json")))
    rows = [r for r in csv.DictReader(open(os.path.join(F(f"trace_c{c}"), "run_kernel_trace.csv"))) if "rsb_step_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by = collections.defaultdict(list)
    for r in rows:
        by[klass(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    n_env, steps_tr = 4096, 20
    out = [f"config {c}: {b['config']['workload'][:120]}...",
           f"  rocprofv3 --kernel-trace over bench.py --no-cpu --no-secondary --config {c} --steps {steps_tr} --warmup 5: launches by kernel class (us)"]
    for k, d in by.items():
        d = np.array(d)
        out.append(f"    {k:34s} {len(d):5d} launches  mean {d.mean():9.1f}  p50 {np.median(d):9.1f}  max {d.max():9.1f}")
    res = [k for k in by if k.startswith("resident")]
    if res:
        d = np.array(by[res[0]])[-7:]       # the timed brackets: the last 7 resident launches (2 untimed ones before them)
        alg = ABYTES[c] * n_env * 4 * steps_tr
        out.append(f"    the 7 timed resident launches ({steps_tr} control steps each): mean {d.mean():.1f} us = {d.mean() / steps_tr:.2f} us per control step -> "
                   f"{alg / 1e6:.1f} MB algorithmic / launch = {alg / d.mean() / 1e3:.1f} GB/s = {100 * alg / d.mean() / 1e3 / 8000:.2f} % of 8 TB/s")
    r = b["roofline"]
    out.append(f"  bench line without a profiler ({'driver flags' if c == 2 else 'default flags'}, steps {b['steps']}): value {b['value'] / 1e6:.1f} M env-steps/s ({b['value_leg']}; 7 brackets {b['value_min'] / 1e6:.1f}-{b['value_max'] / 1e6:.1f}), "
               f"pipelined {(b.get('pipelined') or {}).get('value', 0) / 1e6:.1f} M, lock-step {(b.get('lockstep') or {}).get('value', 0) / 1e6:.1f} M; roofline.kernel_ms_mean {r['kernel_ms_mean'] * 1e3:.1f} us per launch of {r.get('control_steps_per_launch', 1)} control steps, "
               f"achieved {r['achieved']:.1f} GB/s = frac {r['frac']:.5f}")
    f, w = counters(f"pmc_fetch_c{c}"), counters(f"pmc_write_c{c}")
    steps_pmc = 50 if c != 5 else 30
    for k in f:
        fk, wk = np.mean(f[k]["FETCH_SIZE"]), np.mean(w[k]["WRITE_SIZE"])
        hbm = 1024.0 * (2.0 * fk + wk)
        if k.startswith("resident"):
            per = hbm / steps_pmc
            out.append(f"  HBM traffic, {k}: 2 x {fk:.0f} KB + {wk:.0f} KB = {hbm / 1e6:.2f} MB per launch of {steps_pmc} control steps = {per / 1e6:.3f} MB per control step = "
                       f"{per / (ABYTES[c] * n_env):.2f} x the FUSED algorithmic bytes ({ABYTES[c] * n_env / 1e6:.2f} MB), {per / (ABYTES[c] * n_env * 4):.3f} x the unfused contract bytes")
            json.dump({"fetch_kb_per_launch_raw": fk, "write_kb_per_launch_raw": wk, "control_steps_per_launch": steps_pmc, "hbm_bytes_per_launch": hbm, "hbm_bytes_per_control_step": per,
                       "note": "2 x FETCH_SIZE + WRITE_SIZE per resident launch / control steps per launch, calibrated as profiles/r02_traffic_calibration.txt; the obs block is overwritten in place "
                               "(no consumer at N = 1) and stays in the write-back L2 - a [K, N, obs] rollout stride adds 0.80 MB of obs rows per control step",
                       "kernel": k, "workload": f"bench.py --no-cpu --no-secondary --config {c} --steps {steps_pmc} --warmup 10", "round": tag},
                      open(os.path.join(P, f"{tag}_pmc_traffic_resident" + ("" if c == 2 else f"_config{c}") + ".json"), "w"), indent=1)
        elif k.startswith("plain"):
            out.append(f"  HBM traffic, {k} (lock-step leg): 2 x {fk:.0f} KB + {wk:.0f} KB = {hbm / 1e6:.2f} MB per launch = {hbm / (ABYTES[c] * n_env * 4):.2f} x the unfused algorithmic bytes")
            extra = {}
            if c == 2:
                a, bq = counters("pmc_sq"), counters("pmc_sq2")
                extra = {"counters": {**{n: float(np.mean(v)) for n, v in a[k].items()}, **{n: float(np.mean(v)) for n, v in bq[k].items()}}}
            json.dump({**extra, "fetch_kb_per_launch_raw": fk, "write_kb_per_launch_raw": wk, "hbm_bytes_per_launch": hbm,
                       "note": "2 x FETCH_SIZE + WRITE_SIZE per launch, calibrated on known byte counts (profiles/r02_traffic_calibration.txt; same counters, same correction)",
                       "kernel": k, "workload": f"bench.py --no-cpu --no-secondary --config {c} --steps {steps_pmc} --warmup 10 (lock-step leg: 4096 envs x 4 sub-steps per launch)", "round": tag},
                      open(os.path.join(P, f"{tag}_pmc_traffic" + ("" if c == 2 else f"_config{c}") + ".json"), "w"), indent=1)
    if "cpu_baseline" in b:
        out.append(f"  cpu_baseline {b['cpu_baseline']['value'] / 1e6:.2f} M env-steps/s on {b['cpu_baseline']['cores']} threads (single thread {b['cpu_baseline']['single_thread'] / 1e6:.3f} M) -> GPU / CPU = {b['value'] / b['cpu_baseline']['value']:.1f}")
    lines.append("\n".join(out))
txt = f"rocprofv3 --kernel-trace --stats / --pmc over bench.py (MI355X, {tag}; tools/evidence_r06.sh)\n\n" + "\n\n".join(lines) + "\n"
open(os.path.join(P, f"{tag}_rocprofv3_step_kernel_summary.txt"), "w").write(txt)
print(txt)

# ---- SQ counters per kernel class (config 2)
a, bq = counters("pmc_sq"), counters("pmc_sq2")
t = [f"rocprofv3 --pmc <counters> -- python bench.py --no-cpu --no-secondary --steps 50 --warmup 10   (MI355X, {tag}, config 2; separate passes per counter group, means per dispatch)"]
for k in a:
    c = {**{n: float(np.mean(v)) for n, v in a[k].items()}, **{n: float(np.mean(v)) for n, v in bq.get(k, {}).items()}}
    W = 1024
    per_step = 50.0 if k.startswith("resident") else 1.0
    wave_cyc = 4 * c["SQ_WAVE_CYCLES"] / W / per_step
    t.append(f"\n{k}  ({len(a[k]['SQ_WAVE_CYCLES'])} dispatches; {W} single-wave workgroups" + ("; per CONTROL STEP of a 50-step launch" if per_step > 1 else "") + ")")
    t.append(f"  mean wave {wave_cyc / 1e3:.1f}k cycles per control step;  VALU {c['SQ_INSTS_VALU'] / W / per_step / 1e3:.2f}k, SALU {c['SQ_INSTS_SALU'] / W / per_step / 1e3:.2f}k, LDS {c['SQ_INSTS_LDS'] / W / per_step / 1e3:.2f}k instructions per wave and control step")
    t.append(f"  an instruction executing {100 * c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']:.1f} % of wave time; parked at s_waitcnt {100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.1f} %; waiting to issue {100 * c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.1f} %")
    if "SQ_ACTIVE_INST_VALU" in c:
        t.append(f"  VALU busy {100 * c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:.0f} % of wave time; LDS bank conflicts {100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_WAVE_CYCLES']:.1f} %; VMEM instructions {c['SQ_INSTS_VMEM'] / W / per_step:.0f} per wave and control step")
    ninst = c["SQ_INSTS_VALU"] + c["SQ_INSTS_SALU"] + c["SQ_INSTS_LDS"]
    t.append(f"  {wave_cyc * per_step / (ninst / W):.1f} cycles per instruction (VALU + SALU + LDS) on the mean wave")
open(os.path.join(P, f"{tag}_pmc_summary.txt"), "w").write("\n".join(t) + "\n")
print("\n".join(t))
