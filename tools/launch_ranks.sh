#!/bin/bash
# Start N ranks of a program on one node, one per GPU, with the environment the engine's communicator start-up reads
# (RANK / WORLD_SIZE / LOCAL_RANK; torch.distributed.run --no-python exports the same).  Usage:
#   tools/launch_ranks.sh N [--shm] -- dmrg.x_amd/dmrgx-square-lattice -Lx 20 -Ly 8 ...
# --shm: all ranks share GPU 0 through the host-staged back-end (rehearsal on a one-GPU box).
set -u
N=$1; shift
mode=rccl
if [ "${1:-}" = "--shm" ]; then mode=shm; shift; fi
[ "${1:-}" = "--" ] && shift
export WORLD_SIZE=$N DMRGX_COMM=$mode DMRGX_RDZV_FILE=${DMRGX_RDZV_FILE:-/tmp/dmrgx_rdzv_$$} DMRGX_SHM_NAME=${DMRGX_SHM_NAME:-dmrgx_shm_$$}
export HSA_ENABLE_IPC_MODE_LEGACY=0
pids=()
for r in $(seq 0 $((N - 1))); do
  RANK=$r LOCAL_RANK=$r "$@" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
exit $rc
