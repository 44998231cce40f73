#!/bin/bash
# first warm-up steps of 20x8 m=6 with a given library
cp tools/ab/$1 dmrg.x_amd/libdmrgx_hip.so
mkdir -p gpurun_out/r20 && cd gpurun_out/r20 && rm -rf data && mkdir data
timeout -k 5 120 ../../dmrg.x_amd/dmrgx-square-lattice -Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -qn_sector 1 -mwarmup 6 -nsweeps 0 -H_eps_tol 1e-13 -data_dir data/ 2>&1 | grep -E "WARMUP  |dmrgx\]|Energy|E=" | head -8
