"""Turn gpurun_out/<tag>/ (written by tools/collect_profiles.sh on the GPU box) into the tracked files profiles/<tag>_*.
usage: python tools/summarise_profiles.py r01"""
import csv, json, os, shutil, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
O, P = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
for src, dst in [("bench_default.json", "bench_default.json"), ("bench_maxiter30.json", "bench_maxiter30.json"),
                 ("bench_noreset.json", "bench_noreset.json"), ("trace/bench_kernel_stats.csv", "rocprofv3_kernel_stats.csv"),
                 ("trace/bench_domain_stats.csv", "rocprofv3_domain_stats.csv"), ("pmc_traffic.json", "pmc_traffic.json")]:
    shutil.copy(os.path.join(O, src), os.path.join(P, f"{tag}_{dst}"))
rows = [r for r in csv.DictReader(open(os.path.join(O, "trace", "bench_kernel_trace.csv"))) if "rsb_step_kernel" in r["Kernel_Name"]]
d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]) / 1e3
b = json.load(open(os.path.join(O, "bench_default.json")))
steps, km = b["steps"], b["roofline"]["kernel_ms_mean"] * 1e3
r = rows[-1]
txt = f"""rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu      (MI355X, {tag})
kernel: {r['Kernel_Name']}   grid {r['Grid_Size_X']} work-items, workgroup {r['Workgroup_Size_X']} (= one wave64; {int(r['Grid_Size_X']) // 64} workgroups = one per SIMD)
resources: see DESIGN.md section 4 (ISA metadata of the build); trace columns: VGPR_Count {r['VGPR_Count']}, Accum_VGPR_Count {r['Accum_VGPR_Count']}, SGPR_Count {r['SGPR_Count']}, Scratch_Size {r['Scratch_Size']} (dynamic LDS is not shown by the trace)

all {len(d)} launches ({b['warmup']} warm-up control steps from the initial drop + {steps} timed): mean {d.mean():.1f} us  (= AverageNs of {tag}_rocprofv3_kernel_stats.csv)
last {steps} launches = the timed region of bench.py:   mean {d[-steps:].mean():.1f} us   p50 {np.median(d[-steps:]):.1f}   p90 {np.percentile(d[-steps:], 90):.1f}   max {d[-steps:].max():.1f}
first {b['warmup']} launches (free fall, touchdown, settling): mean {d[:b['warmup']].mean():.1f} us   min {d[:b['warmup']].min():.1f}

bench.py without a profiler attached (HIP event ring on the launch stream, {tag}_bench_default.json): kernel_ms_mean {km:.1f} us, p50 {b['roofline']['kernel_ms_p50'] * 1e3:.1f} us
  -> traced timed region vs bench.py: {100 * (d[-steps:].mean() / km - 1):+.1f} %  (bench.py brackets every 8th launch; the bracket includes a few us of event handling,
     which is why it reads above the traced kernel duration and above ms_per_step)

one launch = {b['config']['envs_per_gpu']} envs x {b['config']['substeps_per_step']} sub-steps = {b['config']['envs_per_gpu'] * b['config']['substeps_per_step']} env-steps; algorithmic bytes 456 B x that = {b['roofline']['algorithmic_bytes_per_launch'] / 1e6:.2f} MB
roofline.achieved = {b['roofline']['algorithmic_bytes_per_launch'] / 1e6:.2f} MB / {km:.1f} us = {b['roofline']['achieved']:.1f} GB/s = {100 * b['roofline']['frac']:.2f} % of 8 TB/s  (latency bound, see {tag}_pmc_summary.txt)
value = {b['value'] / 1e6:.1f} M env-steps/s, {b['ms_per_step']:.4f} ms per control step (one launch per control step)
cpu_baseline = {b['cpu_baseline']['value'] / 1e6:.2f} M env-steps/s on {b['cpu_baseline']['cores']} threads (single thread {b['cpu_baseline']['single_thread'] / 1e6:.3f} M)
"""
open(os.path.join(P, f"{tag}_rocprofv3_step_kernel_summary.txt"), "w").write(txt)
c = json.load(open(os.path.join(O, "pmc_traffic.json")))["counters"]
W = int(r["Grid_Size_X"]) // 64
ninst = c["SQ_INSTS_VALU"] + c["SQ_INSTS_SALU"] + c["SQ_INSTS_LDS"]
t = f"""rocprofv3 --pmc <counters> --output-format csv -- python bench.py --no-cpu --steps 50 --warmup 50     (MI355X, {tag})
Separate passes per counter group (never combined with trace domains); means per dispatch of
{c['kernel']} over 100 dispatches ({W} single-wave workgroups). Raw per-dispatch means: {tag}_pmc_traffic.json.

pass 1  FETCH_SIZE            {c['FETCH_SIZE']:12.1f}   KB  (= TCC_EA0_RDREQ x 64 B; gfx950 under-reports WIDE 16 B/lane streams by 2x; this kernel's
                                             row accesses are 4 B/lane: factor uncalibrated, MI355X_MICROARCH.md HBM section)
pass 2  WRITE_SIZE            {c['WRITE_SIZE']:12.1f}   KB  (uncalibrated)
        => memory-side traffic per launch ~ {(c['FETCH_SIZE'] + c['WRITE_SIZE']) / 1024:.2f} MB raw.  Needed by the fused control step: 4096 x (state 37 + targets 55
           floats read; state 37 + obs 49 floats + 8 x 48 B contacts + 12 B written) ~ 4.1 MB; SURVEY 8d contract number (unfused) 7.47 MB.
           Nothing is re-read: state, targets and model tables are loaded once per launch and stay in LDS for the 4 sub-steps.
pass 3  SQ_WAVE_CYCLES        {c['SQ_WAVE_CYCLES']:12.0f}   quad-cycles over all waves: {4 * c['SQ_WAVE_CYCLES'] / W / 1e3:.0f}k cycles per wave per launch (mean wave)
        SQ_BUSY_CYCLES        {c['SQ_BUSY_CYCLES']:12.0f}
        SQ_ACTIVE_INST_ANY    {c['SQ_ACTIVE_INST_ANY']:12.0f}   {100 * c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']:.1f} % of wave time an instruction is executing
        SQ_WAIT_ANY           {c['SQ_WAIT_ANY']:12.0f}   {100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.1f} % parked at s_waitcnt (LDS round trips, nothing else to run)
        SQ_WAIT_INST_ANY      {c['SQ_WAIT_INST_ANY']:12.0f}   {100 * c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.1f} % waiting for an instruction to issue (branches / fetch)
        SQ_INSTS_VALU         {c['SQ_INSTS_VALU']:12.0f}   {c['SQ_INSTS_VALU'] / W / 1e3:.1f}k VALU instructions per wave per launch = {c['SQ_INSTS_VALU'] / W / 4e3:.1f}k per sub-step
        SQ_INSTS_SALU         {c['SQ_INSTS_SALU']:12.0f}   {c['SQ_INSTS_SALU'] / W / 1e3:.1f}k per wave
        SQ_INSTS_LDS          {c['SQ_INSTS_LDS']:12.0f}   {c['SQ_INSTS_LDS'] / W / 1e3:.2f}k per wave
pass 4  SQ_ACTIVE_INST_VALU   {c['SQ_ACTIVE_INST_VALU']:12.0f}   {c['SQ_ACTIVE_INST_VALU'] / c['SQ_INSTS_VALU']:.2f} quad-cycles per VALU instruction; VALU busy {100 * c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:.0f} % of wave time
        SQ_ACTIVE_INST_LDS    {c['SQ_ACTIVE_INST_LDS']:12.0f}
        SQ_ACTIVE_INST_SCA    {c['SQ_ACTIVE_INST_SCA']:12.0f}
        SQ_WAIT_INST_LDS      {c['SQ_WAIT_INST_LDS']:12.0f}
        SQ_LDS_BANK_CONFLICT  {c['SQ_LDS_BANK_CONFLICT']:12.0f}   {100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_WAVE_CYCLES']:.1f} % of wave time
        SQ_INSTS_VMEM         {c['SQ_INSTS_VMEM']:12.0f}   {c['SQ_INSTS_VMEM'] / W:.0f} per wave (state / target rows in; state, obs, contacts out)
        GRBM_GUI_ACTIVE       {c['GRBM_GUI_ACTIVE']:12.0f}

Reading: latency bound, neither memory nor MFMA bound.  4096 envs = {W} waves = one wave per SIMD (430 registers ->
occupancy 1, and there are no further envs to make a second wave), so every s_waitcnt and every issue bubble is
exposed: {4 * c['SQ_WAVE_CYCLES'] / ninst:.1f} cycles per instruction on average.  HBM traffic ~{(c['FETCH_SIZE'] + c['WRITE_SIZE']) / 1024:.1f} MB per ~{km / 1e3:.2f} ms launch (~{(c['FETCH_SIZE'] + c['WRITE_SIZE']) / 1024 / km * 1e3:.0f} GB/s).
The launch ends with its slowest wave (the env with the longest Gauss-Seidel solves), ~1.7x the median wave.
"""
open(os.path.join(P, f"{tag}_pmc_summary.txt"), "w").write(t)
print(txt); print(t[-700:])
