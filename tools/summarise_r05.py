"""Turn gpurun_out/<tag>/ (written on the GPU box by the evidence call of tools/lease.sh, steps labelled as in EVIDENCE below) into the tracked
files profiles/<tag>_*.   usage: python tools/summarise_r05.py r05
EVIDENCE (the labels this script looks for; `bash tools/lease.sh r05 ...`, printed by --steps):
  pytest@pytest smoke@smoke bench@default20 bench@default300 bench@c2 bench@c3 bench@c5 bench@c2_lockstep bench@closed20 bench@closed300
  trace@trace_c2 trace@trace_c3 trace@trace_c5 trace@trace_closed  pmc@pmc_fetch_cN / pmc_write_cN (N = 2, 3, 5)  pmc@pmc_sq pmc@pmc_sq2 pmc@pmc_sq_c5
Register counts are read from the ISA this tree compiles to (hipcc -S of the benchmark's instances, .vgpr_count / NumVgprs /
NumAgprs / spill counts): rocprofv3's VGPR_Count column reports the arch-VGPR allocation granule, not arch + accumulation file."""
import collections, csv, glob, json, os, re, shutil, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
O, P = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")


def F(label, ext=""):
    """the file / directory the step labelled `label` wrote (lease.sh names them NN_<label><ext>)"""
    hits = sorted(glob.glob(os.path.join(O, f"[0-9][0-9]_{label}{ext}")))
    if not hits:
        raise SystemExit(f"no output of a step labelled {label!r} under {O}")
    return hits[-1]


cp = lambda src, dst: shutil.copy(src, os.path.join(P, f"{tag}_{dst}"))
for c, name in ((2, "config2"), (3, "config3"), (5, "config5")):
    cp(F(f"c{c}", ".json"), f"bench_{name}.json")
    cp(os.path.join(F(f"trace_c{c}"), "run_kernel_stats.csv"), f"rocprofv3_kernel_stats_{name}.csv")
cp(F("default20", ".json"), "bench_default_line_steps20_warmup5.json")
cp(F("default300", ".json"), "bench_default_line.json")
cp(F("closed20", ".json"), "bench_closed_loop_steps20_warmup5.json")
cp(F("closed300", ".json"), "bench_closed_loop.json")
cp(os.path.join(F("trace_closed"), "run_kernel_stats.csv"), "rocprofv3_kernel_stats_closed_loop.csv")


def isa_resources(lpe, kmax, ml):
    """register file use of one kernel instance, from the ISA (compiled here; hipcc cross-compiles without a GPU)"""
    sys.path.insert(0, ROOT)
    from raisimlib_amd import build as rb
    csrc = os.path.join(ROOT, "raisimlib_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["hipcc", *rb.FLAGS, "-I", os.path.join(ROOT, "include"), "-I", csrc, f"-DRSB_I_LPE={lpe}", f"-DRSB_I_KMAX={kmax}", "-DRSB_I_CL=0",
                        f"-DRSB_I_ML={ml}", "-DRSB_I_PROF=0", "--cuda-device-only", "-S", "-o", out, os.path.join(csrc, "step_instance.hip")],
                       check=True, capture_output=True)
        t = open(out).read()
    g = lambda k: int(re.search(r"\.%s:\s*(\d+)" % k, t).group(1))
    body = t[t.index(":\n", t.index("rsb_step_kernel")):]
    body = body[:body.index("s_endpgm")]
    ins = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
    cnt = lambda pat: sum(1 for l in ins if re.match(pat, l))
    nv, na = re.search(r"NumVgprs: (\d+)", t).group(1), re.search(r"NumAgprs: (\d+)", t).group(1)
    return ("ISA: NumVgprs %s + NumAgprs %s = %d of 512, SGPR %d, sgpr_spill_count %d, vgpr_spill_count %d, scratch %d B; %d instructions "
            "(%d VALU of which %d v_accvgpr moves and %d v_readlane/v_writelane, %d LDS, %d s_waitcnt, %d v_mfma)" % (
                nv, na, g("vgpr_count"), g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), len(ins),
                cnt(r"v_"), cnt(r"v_accvgpr"), cnt(r"v_(read|write)lane"), cnt(r"ds_"), cnt(r"s_waitcnt"), cnt(r"v_mfma")))


ISA = {2: isa_resources(16, 8, 4), 3: isa_resources(16, 8, 4), 5: isa_resources(32, 16, 12)}
open(os.path.join(P, f"{tag}_pytest_gpu_tail.txt"), "w").write("".join(open(F("pytest", ".log")).readlines()[-16:]) + open(F("smoke", ".log")).read())


def counters(sub):
    fs = glob.glob(os.path.join(F(sub), "**", "*counter_collection.csv"), recursive=True)
    acc, n, kn = collections.defaultdict(float), collections.Counter(), ""
    for r in csv.DictReader(open(fs[0])):
        if "rsb_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1; kn = r["Kernel_Name"]
    return {k: v / n[k] for k, v in acc.items()}, kn


lines = []
for c, name, abytes in ((2, "config2", 456.0), (3, "config3", 520.0), (5, "config5", 1080.0)):
    b = json.load(open(F(f"c{c}", ".json")))
    rows = [r for r in csv.DictReader(open(os.path.join(F(f"trace_c{c}"), "run_kernel_trace.csv"))) if "rsb_step_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]) / 1e3
    steps, km = b["steps"], b["roofline"]["kernel_ms_mean"] * 1e3
    pipe_note, pre = "", 0
    if b.get("lockstep"):
        # round 4: the trace holds, in order, pre-roll + warm-up (pipelined kernel class, joined after every step), the TIMED REGION (pipelined,
        # overlapping), the sampling pass (pipelined class, one launch at a time), the lock-step leg (the plain kernel class)
        pre = b["config"]["preroll_control_steps"] + b["warmup"]
        start = np.array([int(r["Start_Timestamp"]) for r in rows]); end = np.array([int(r["End_Timestamp"]) for r in rows])
        tr = slice(pre, pre + steps)
        span = (end[tr].max() - start[tr].min()) / 1e3
        overlap = float(np.mean(start[pre + 1:pre + steps] < end[pre:pre + steps - 1]))
        pipe_note = (f"\n  pipelined timed region ({steps} launches of the | 16 kernel class): start -> end mean {d[tr].mean():.1f} us, p50 {np.median(d[tr]):.1f}, max {d[tr].max():.1f};"
                     f" first start -> last end {span:.1f} us = {span / steps:.2f} us per launch; {100 * overlap:.0f} % of the launches start before their predecessor has ended"
                     f"\n  lock-step leg (last {steps} launches, plain kernel class): mean {d[-steps:].mean():.1f} us, p50 {np.median(d[-steps:]):.1f}, max {d[-steps:].max():.1f}"
                     f"   (bench.py lockstep: {b['lockstep']['ms_per_step'] * 1e3:.1f} us per control step, {b['lockstep']['value'] / 1e6:.1f} M env-steps/s)")
        d_timed = d[tr]
    else:
        d_timed = d[-steps:]
    f, kn = counters(f"pmc_fetch_c{c}"); w, _ = counters(f"pmc_write_c{c}")
    fetch_kb, write_kb = f["FETCH_SIZE"], w["WRITE_SIZE"]
    hbm = 1024.0 * (2.0 * fetch_kb + write_kb)
    extra = {}
    if c == 2:
        ca, _ = counters("pmc_sq"); cb, _ = counters("pmc_sq2")
        extra = {"counters": {**ca, **cb}}      # bench.py derives roofline.valu_issue from SQ_INSTS_VALU of this pass
    json.dump({**extra, "fetch_kb_per_launch_raw": fetch_kb, "write_kb_per_launch_raw": write_kb, "hbm_bytes_per_launch": hbm,
               "note": "2 x FETCH_SIZE + WRITE_SIZE per launch, calibrated on known byte counts (profiles/r02_traffic_calibration.txt; same counters, same correction)",
               "kernel": kn, "workload": f"bench.py --no-cpu --no-secondary --config {c} --steps 50 --warmup 50 (4096 envs x 4 sub-steps per launch)", "round": tag},
              open(os.path.join(P, f"{tag}_pmc_traffic" + ("" if c == 2 else f"_config{c}") + ".json"), "w"), indent=1)
    r = rows[-1]
    alg = abytes * b["config"]["envs_per_gpu"] * b["config"]["substeps_per_step"]
    lines.append(f"""config {c}: {b['config']['workload'][:110]}...
  kernel {(rows[pre]['Kernel_Name'][:62] + ' (pipelined twin; timed region) | ') if b.get('lockstep') else ''}{r['Kernel_Name'][:70]}  grid {r['Grid_Size_X']} work-items = {int(r['Grid_Size_X']) // 64} single-wave workgroups
  {ISA[c]}
  rocprofv3 --kernel-trace: timed region's {steps} launches mean {d_timed.mean():.1f} us  p50 {np.median(d_timed):.1f}  p90 {np.percentile(d_timed, 90):.1f}  max {d_timed.max():.1f}   (all {len(d)} launches incl. pre-roll: mean {d.mean():.1f} us){pipe_note}
  bench.py without a profiler: value {b['value'] / 1e6:.1f} M env-steps/s, {b['ms_per_step']:.4f} ms per control step; HIP-event brackets of {b['roofline']['kernel_launches_timed']} launches: mean {km:.1f} us (trace vs bench: {100 * (d_timed.mean() / km - 1):+.1f} %)
  algorithmic bytes {abytes:.0f} B x {b['config']['envs_per_gpu'] * b['config']['substeps_per_step']} env-steps = {alg / 1e6:.2f} MB per launch -> roofline.achieved {b['roofline']['achieved']:.1f} GB/s = {100 * b['roofline']['frac']:.2f} % of 8 TB/s
  HBM traffic (PMC, calibrated): 2 x {fetch_kb:.0f} KB + {write_kb:.0f} KB = {hbm / 1e6:.2f} MB per launch = {hbm / alg:.2f} x the unfused algorithmic bytes""")
    if "cpu_baseline" in b:
        lines.append(f"  cpu_baseline {b['cpu_baseline']['value'] / 1e6:.2f} M env-steps/s on {b['cpu_baseline']['cores']} threads (single thread {b['cpu_baseline']['single_thread'] / 1e6:.3f} M) -> GPU / CPU = {b['value'] / b['cpu_baseline']['value']:.1f}")
open(os.path.join(P, f"{tag}_rocprofv3_step_kernel_summary.txt"), "w").write(
    f"rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu --config C     (MI355X, {tag}; tools/lease.sh, steps trace@trace_cN)\n\n" + "\n\n".join(lines) + "\n")
print("\n\n".join(lines))

c1, kn = counters("pmc_sq"); c2, _ = counters("pmc_sq2"); c = {**c1, **c2}
b = json.load(open(F("c2", ".json")))
W = b["config"]["envs_per_gpu"] * b["config"]["lanes_per_env"] // 64
km = b["roofline"]["kernel_ms_mean"] * 1e3
km = json.load(open(F("c2_lockstep", ".json")))["roofline"]["kernel_ms_mean"] * 1e3     # (the SQ passes run in lock-step: the plain kernel class and its own launch time)
ninst = c["SQ_INSTS_VALU"] + c["SQ_INSTS_SALU"] + c["SQ_INSTS_LDS"]
wave_cyc = 4 * c["SQ_WAVE_CYCLES"] / W
t = f"""rocprofv3 --pmc <counters> --output-format csv -- python bench.py --no-cpu --steps 50 --warmup 50     (MI355X, {tag}, config 2)
Separate passes per counter group (never combined with trace domains); means per dispatch of {kn[:60]} ({W} single-wave workgroups).

pass A  SQ_WAVE_CYCLES        {c['SQ_WAVE_CYCLES']:12.0f}   quad-cycles over all waves: {wave_cyc / 1e3:.0f}k cycles per wave per launch (MEAN wave; the launch lasts as long as its slowest wave: {km:.1f} us)
        SQ_BUSY_CYCLES        {c['SQ_BUSY_CYCLES']:12.0f}
        SQ_ACTIVE_INST_ANY    {c['SQ_ACTIVE_INST_ANY']:12.0f}   {100 * c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']:.1f} % of wave time an instruction is executing
        SQ_WAIT_ANY           {c['SQ_WAIT_ANY']:12.0f}   {100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.1f} % parked at s_waitcnt
        SQ_WAIT_INST_ANY      {c['SQ_WAIT_INST_ANY']:12.0f}   {100 * c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.1f} % waiting for an instruction to issue (branches / fetch)
        SQ_INSTS_VALU         {c['SQ_INSTS_VALU']:12.0f}   {c['SQ_INSTS_VALU'] / W / 1e3:.1f}k VALU instructions per wave per launch = {c['SQ_INSTS_VALU'] / W / 4e3:.1f}k per sub-step
        SQ_INSTS_SALU         {c['SQ_INSTS_SALU']:12.0f}   {c['SQ_INSTS_SALU'] / W / 1e3:.1f}k per wave
        SQ_INSTS_LDS          {c['SQ_INSTS_LDS']:12.0f}   {c['SQ_INSTS_LDS'] / W / 1e3:.2f}k per wave
pass B  SQ_ACTIVE_INST_VALU   {c['SQ_ACTIVE_INST_VALU']:12.0f}   {c['SQ_ACTIVE_INST_VALU'] / c['SQ_INSTS_VALU']:.2f} quad-cycles per VALU instruction; VALU busy {100 * c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:.0f} % of (mean) wave time
        SQ_ACTIVE_INST_LDS    {c['SQ_ACTIVE_INST_LDS']:12.0f}
        SQ_ACTIVE_INST_SCA    {c['SQ_ACTIVE_INST_SCA']:12.0f}
        SQ_WAIT_INST_LDS      {c['SQ_WAIT_INST_LDS']:12.0f}
        SQ_LDS_BANK_CONFLICT  {c['SQ_LDS_BANK_CONFLICT']:12.0f}   {100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_WAVE_CYCLES']:.1f} % of wave time
        SQ_INSTS_VMEM         {c['SQ_INSTS_VMEM']:12.0f}   {c['SQ_INSTS_VMEM'] / W:.0f} per wave
        GRBM_GUI_ACTIVE       {c['GRBM_GUI_ACTIVE']:12.0f}

VALU issue-slot use over the LAUNCH (what roofline.valu_issue reports): {c['SQ_INSTS_VALU'] / W:.0f} VALU x 4 cycles / ({km:.1f} us x clock) - see the bench line;
over the mean wave: {100 * 4 * c['SQ_INSTS_VALU'] / W / wave_cyc:.0f} %.  {wave_cyc / (ninst / W):.1f} cycles per instruction (VALU + SALU + LDS) on the mean wave.
"""
open(os.path.join(P, f"{tag}_pmc_summary.txt"), "w").write(t)
json.dump({"counters": c, "kernel": kn, "waves": W}, open(os.path.join(P, f"{tag}_pmc_counters.json"), "w"), indent=1)
print(t)
c5, kn5 = counters("pmc_sq_c5")
b5 = json.load(open(F("c5", ".json")))
W5 = b5["config"]["envs_per_gpu"] * b5["config"]["lanes_per_env"] // 64
open(os.path.join(P, f"{tag}_pmc_summary_config5.txt"), "w").write(
    f"config 5 (Atlas-like), {kn5[:60]}, {W5} waves, means per dispatch:\n" + "\n".join(f"  {k:22s} {v:14.0f}" for k, v in c5.items()) +
    f"\n  VALU instructions per wave per launch {c5['SQ_INSTS_VALU'] / W5 / 1e3:.1f}k; mean wave {4 * c5['SQ_WAVE_CYCLES'] / W5 / 1e3:.0f}k cycles; VALU busy {100 * c5['SQ_ACTIVE_INST_VALU'] / c5['SQ_WAVE_CYCLES']:.0f} % of wave time; s_waitcnt {100 * c5['SQ_WAIT_ANY'] / c5['SQ_WAVE_CYCLES']:.0f} %\n")

# ---- the closed loop's launches in the kernel trace (trace@trace_closed: bench.py --closed-loop-only)
rows = list(csv.DictReader(open(os.path.join(F("trace_closed"), "run_kernel_trace.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cl = json.load(open(F("closed300", ".json")))["closed_loop"]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
txt = []
for kind, key, blk in (("linear policy 12 x 34", "linear_stage_kernel", cl), ("actor network 34-128-128-12", "mlp_stage_kernel", cl.get("mlp"))):
    if not blk or not blk.get("pipelined"):
        continue
    stage = [r for r in rows if key in r["Kernel_Name"]]
    pers = [r for r in stage if dur(r) > 400]
    short = [dur(r) for r in stage if dur(r) <= 400]
    txt += [f"closed loop, stage = {kind} (bench.py --closed-loop-only, {tag}): pipelined {blk['pipelined']['value'] / 1e6:.1f} M env-steps/s ({blk['pipelined']['ms_per_step'] * 1e3:.1f} us per control step), lock-step {blk['lockstep']['value'] / 1e6:.1f} M",
            f"  kernel trace of the same command: {len(pers)} persistent launches of the action stage (one per run; {', '.join(f'{dur(r) / 1e3:.2f} ms' for r in pers)}), {len(short)} per-pass launches of the lock-step leg (mean {np.mean(short) if short else 0:.1f} us)"]
    if pers:
        p = max(pers, key=dur)
        s0, e0 = int(p["Start_Timestamp"]), int(p["End_Timestamp"])
        st = [r for r in rows if "rsb_step_kernel" in r["Kernel_Name"] and int(r["Start_Timestamp"]) >= s0 and int(r["End_Timestamp"]) <= e0 + 1000]
        d = np.array([dur(r) for r in st]); starts = np.array([int(r["Start_Timestamp"]) for r in st]); ends = np.array([int(r["End_Timestamp"]) for r in st])
        txt.append(f"  longest run: {len(st)} step launches inside the stage's {dur(p) / 1e3:.2f} ms; start -> end mean {d.mean():.1f} us, one completes every {(ends.max() - starts.min()) / 1e3 / len(st):.1f} us; "
                   f"{100 * np.mean(starts[1:] < ends[:-1]):.0f} % start before their predecessor has ended")
open(os.path.join(P, f"{tag}_closed_loop_trace_summary.txt"), "w").write("\n".join(txt) + "\n")
print("\n".join(txt))
