#!/bin/bash
# A/B of BUILDS and OPTIONS on one box: bash tools/ab_builds.sh ROUNDS label:variant[:options] ...   (libmvsim_<variant>.so built beforehand;
# options = MVSIM_OPTIONS string).  Prints the bench line's stage times per run, then instruction and wait counters of the sampler's kernels.
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ROUNDS=$1; shift
cp multiview-simulation_amd/libmvsim.so gpurun_out/libmvsim_keep.so
for r in $(seq $ROUNDS); do
  for spec in "$@"; do
    IFS=: read label variant opts <<< "$spec"
    cp multiview-simulation_amd/libmvsim_$variant.so multiview-simulation_amd/libmvsim.so
    MVSIM_OPTIONS="$opts" python bench.py --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams --no-compact-queue-leg --no-small-views --no-main-iteration > gpurun_out/ab_lib_$label.log 2>&1 || { tail -5 gpurun_out/ab_lib_$label.log; exit 1; }
    python - "$label" gpurun_out/ab_lib_$label.log <<'PY'
import json, sys
for l in open(sys.argv[2]):
    if l.startswith("{"):
        d = json.loads(l); s = d["roofline"]["stage_ms"]; ne = d["no_empty_space"]
        print(f"[{sys.argv[1]}] {d['value']:.0f} Mvox/s dense {d['value_dense']:.0f}  total {s['total_ms']:.3f}  rot {s['rotate_ms']:.3f} (dense {ne['rotate_attenuate_ms']:.3f}) conv {s['convolve_ms']:.3f} (B {s['pass_b_ms']:.3f} C {s['pass_c_ms']:.3f} D {s['pass_d_ms']:.3f} E {s['pass_e_ms']:.3f}) extract {s['extract_ms']:.3f} (dense {ne['extract_ms']:.3f})", flush=True)
PY
  done
done
if [ -n "$AB_PMC" ]; then
B="bench.py --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams --no-dense-leg --no-main-iteration --no-small-views --no-compact-queue-leg --serial --steps 1 --warmup 1"
for spec in "$@"; do
  IFS=: read label variant opts <<< "$spec"
  cp multiview-simulation_amd/libmvsim_$variant.so multiview-simulation_amd/libmvsim.so
  echo "[$label]"
  export MVSIM_OPTIONS="$opts"
  rm -rf gpurun_out/pi && mkdir -p gpurun_out/pi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/pi -o run -- python3 $B > gpurun_out/pi.log 2>&1
  python3 tools/pmc_insts.py gpurun_out/pi | grep -i "kernel\|$AB_PMC"
  rm -rf gpurun_out/pi && mkdir -p gpurun_out/pi
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d gpurun_out/pi -o run -- python3 $B > gpurun_out/pi.log 2>&1
  python3 tools/pmc_sq_report.py gpurun_out/pi 2>/dev/null | grep -i "kernel\|$AB_PMC" || true
  rm -rf gpurun_out/pi
  unset MVSIM_OPTIONS
done
fi
cp gpurun_out/libmvsim_keep.so multiview-simulation_amd/libmvsim.so
