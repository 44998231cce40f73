#!/bin/bash
# Diagnostic: per-workgroup milestone stamps of the grouped GEMM inside one superblock MatMult (cfg4real by default).
# Builds a SEPARATE library with -DDMRGX_TILE_TRACE (tools/trace/libdmrgx_hip.so; the product library never contains the stamps),
# runs tools/tile_trace.py against it on the GPU box and leaves gpurun_out/trace/{stamps.npy,plan.txt,summary.txt}.
#   here:     tools/tile_trace.sh build
#   GPU box:  tools/tile_trace.sh run [workload]
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  make -s all
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Idmrg.x_amd/csrc -Wall -Wno-unused-function -Wno-pass-failed -Wno-inline-asm -DDMRGX_TILE_TRACE -c dmrg.x_amd/csrc/ggemm.hip -o tools/trace/ggemm.o
  objs=$(ls dmrg.x_amd/csrc/*.o | grep -v ggemm.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/trace/ggemm.o -ldl -lpthread -lrt -o tools/trace/libdmrgx_hip.so
  echo built tools/trace/libdmrgx_hip.so
else
  mkdir -p gpurun_out/trace
  # $3: directory under tools/ holding the instrumented library (default "trace"); $4: output directory
  out=${4:-gpurun_out/trace}; mkdir -p $out
  DMRGX_PLAN_DUMP=$out/plan.txt python3 tools/tile_trace.py ${2:-cfg4real} $out ${3:-trace}
  python3 tools/tile_trace_report.py $out | tee $out/summary.txt
fi
