#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box through gpurun).  Usage: tools/profile.sh <tag> [bench args...]
# Pass 1: kernel trace + stats (per-kernel average duration).  Passes 2..: PMC counters, each in its own run and
# never combined with trace domains other than --kernel-trace (pool rule).
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --no-cpu-baseline --no-sweep "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  NAME=$(echo $PMC | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$NAME -o pmc -- python3 bench.py --no-cpu-baseline --no-sweep --steps 16 --warmup 4 "$@" > $OUT/bench_pmc_$NAME.json 2> $OUT/pmc_$NAME.err || echo "pmc pass $NAME failed"
done
# calibration of FETCH_SIZE / WRITE_SIZE for this kernel's access pattern (8-byte loads per lane): the probe streams a known
# byte count (every group its own operands, 4.3 GB > Infinity Cache) through the same kernel
for PMC in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_calib_$PMC -o pmc -- tools/ggemm_probe 1024 2048 0 0 3 > $OUT/calib_$PMC.out 2> $OUT/calib_$PMC.err || echo "calibration pass $PMC failed"
done
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
