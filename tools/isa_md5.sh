#!/bin/bash
# tools/isa_md5.sh <outdir> <lpe,kmax,cl,ml> ... - device-only assembly of step-kernel instances (build.py's flags) and the md5 of each one's instruction
# stream (labels, comments and directives stripped), for "this refactor leaves the benchmark's classes instruction-for-instruction what they were" checks
O=${1:?outdir}; shift
mkdir -p "$O"
R=$(cd "$(dirname "$0")/.." && pwd)
for inst in "$@"; do
  IFS=, read lpe kmax cl ml <<< "$inst"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt -I $R/include -I $R/raisimlib_amd/csrc \
      -DRSB_I_LPE=$lpe -DRSB_I_KMAX=$kmax -DRSB_I_CL=$cl -DRSB_I_ML=$ml -DRSB_I_PROF=0 --cuda-device-only -S -o "$O/step_${lpe}_${kmax}_${cl}_${ml}.s" $R/raisimlib_amd/csrc/step_instance.hip 2>/dev/null
    grep -E '^\s+[a-z_0-9]+ ' "$O/step_${lpe}_${kmax}_${cl}_${ml}.s" | grep -vE '^\s+\.' | md5sum | awk -v n="$inst" '{print n, $1}' > "$O/step_${lpe}_${kmax}_${cl}_${ml}.md5" ) &
done
wait
cat "$O"/*.md5
