#!/bin/bash
# SQ wait / issue counters of one view at an arbitrary geometry: bash tools/sq_view.sh 1024 1024 1024 31 31 63 4 [options...]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/sqv; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $O/sq -o run -- python3 tools/view_time.py "$@" > $O/sq.log 2>&1
python3 tools/pmc_sq_report.py $O/sq
