#!/bin/bash
# round 6: the resolver as shipped in round 5 (libmvsim_old.so: the loop the compiler had split into items x retries) against the
# single-latch loop (libmvsim_new.so) with 1 / 2 / 4 segments per block, builds A/B on one box: bench stage times, then instruction
# and wait counters of the sampler's two kernels
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
cp multiview-simulation_amd/libmvsim.so gpurun_out/libmvsim_keep.so
run() {   # $1 label, $2 library variant, $3 MVSIM_OPTIONS
  cp multiview-simulation_amd/libmvsim_$2.so multiview-simulation_amd/libmvsim.so
  MVSIM_OPTIONS="$3" python bench.py --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams --no-compact-queue-leg --no-small-views --no-main-iteration > gpurun_out/ab_lib_$1.log 2>&1 || { tail -5 gpurun_out/ab_lib_$1.log; exit 1; }
  python - "$1" gpurun_out/ab_lib_$1.log <<'PY'
import json, sys
for l in open(sys.argv[2]):
    if l.startswith("{"):
        d = json.loads(l); s = d["roofline"]["stage_ms"]
        print(f"[{sys.argv[1]}] {d['value']:.0f} Mvox/s dense {d['value_dense']:.0f}  total {s['total_ms']:.3f}  rot {s['rotate_ms']:.3f} conv {s['convolve_ms']:.3f} extract {s['extract_ms']:.3f} (dense {d['no_empty_space']['extract_ms']:.3f})")
PY
}
for r in 1 2; do
  run old old ""
  run new_g1 new "poisson_resolve_group=1"
  run new_g2 new "poisson_resolve_group=2"
  run new_g4 new "poisson_resolve_group=4"
done
B="bench.py --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams --no-dense-leg --no-main-iteration --no-small-views --no-compact-queue-leg --serial --steps 1 --warmup 1"
pmc() {   # $1 label, $2 variant, $3 options
  cp multiview-simulation_amd/libmvsim_$2.so multiview-simulation_amd/libmvsim.so
  echo "[$1]"
  export MVSIM_OPTIONS="$3"
  rm -rf gpurun_out/pi && mkdir -p gpurun_out/pi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/pi -o run -- python3 $B > gpurun_out/pi.log 2>&1
  python3 tools/pmc_insts.py gpurun_out/pi | grep -i "kernel\|resolve"
  rm -rf gpurun_out/pi && mkdir -p gpurun_out/pi
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d gpurun_out/pi -o run -- python3 $B > gpurun_out/pi.log 2>&1
  python3 tools/pmc_sq_report.py gpurun_out/pi 2>/dev/null | grep -i "kernel\|resolve" || true
  rm -rf gpurun_out/pi
  unset MVSIM_OPTIONS
}
pmc old old ""
pmc new_g1 new "poisson_resolve_group=1"
pmc new_g4 new "poisson_resolve_group=4"
cp gpurun_out/libmvsim_keep.so multiview-simulation_amd/libmvsim.so
