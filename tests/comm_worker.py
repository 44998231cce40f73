"""Worker of the native-communicator tests (csrc/comm.hip through dmrgx_amd.superblock.Communicator).

  comm_worker.py rccl1                 one rank, RCCL back-end: librccl.so is found, ncclCommInitRank(world 1) works and the in-place
                                       all-gather / all-reduce / broadcast call forms run on the device
  comm_worker.py shm RANK WORLD NAME   WORLD ranks share cuda:0 through the host-staged back-end: semantics of every collective, then
                                       the striped eigensolve with the solver's own collectives (opts.comm) against the one-rank solve
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package

load_package()
from dmrgx_amd.superblock import KronPlan, Communicator
from dmrgx_amd.workloads import synthetic_superblock

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
mode = sys.argv[1]

if mode == "rccl1":
    comm = Communicator(0, 1, unique_id=Communicator.unique_id())
    full = torch.arange(4096, dtype=torch.float64, device=dev)
    want = full.clone()
    comm.allgather(full, 4096)
    few = torch.tensor([1.5, -2.0, 3.25], dtype=torch.float64, device=dev)
    comm.allreduce_sum(few)
    comm.bcast(few, 0)
    got = comm.allgather_host(np.array([7.0, 8.0]))
    comm.barrier()
    assert torch.equal(full, want) and few.tolist() == [1.5, -2.0, 3.25] and got.tolist() == [[7.0, 8.0]]
    sb = synthetic_superblock("cfg2", m=64, Ly=3, seed=5)
    plan = KronPlan(sb)
    e_a, _, st_a = plan.eigs_lowest(tol=1e-11, seed=11, comm=comm)          # a communicator on an unstriped plan is simply unused
    e_b, _, _ = plan.eigs_lowest(tol=1e-11, seed=11)
    assert e_a == e_b and st_a.converged == 1
    comm.destroy()
    print("native rccl communicator ok")
    sys.exit(0)

rank, world, name = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
comm = Communicator(rank, world, host_staged_name=name)
# ---- collective semantics -----------------------------------------------------------------------------------------------
seg = 1000
full = torch.zeros(seg * world, dtype=torch.float64, device=dev)
full[rank * seg:(rank + 1) * seg] = torch.arange(seg, dtype=torch.float64, device=dev) + 10000.0 * rank
comm.allgather(full, seg)
want = torch.cat([torch.arange(seg, dtype=torch.float64) + 10000.0 * r for r in range(world)])
assert torch.equal(full.cpu(), want)
v = torch.tensor([1.0 + rank, 0.25 * rank, -3.0], dtype=torch.float64, device=dev)
comm.allreduce_sum(v)
assert v.tolist() == [sum(1.0 + r for r in range(world)), sum(0.25 * r for r in range(world)), -3.0 * world]
b = torch.full((50000,), float(rank), dtype=torch.float64, device=dev)
comm.bcast(b, world - 1)
assert float(b.min()) == float(b.max()) == float(world - 1)
h = comm.allgather_host(np.array([rank, rank * rank], dtype=np.int64))
assert h.tolist() == [[r, r * r] for r in range(world)]
comm.barrier()
# ---- striped eigensolve with the solver's own collectives ---------------------------------------------------------------------
sb = synthetic_superblock("cfg2", m=64, Ly=3, seed=5)
plan = KronPlan(sb, world_size=world, rank=rank)
info = plan.info
e0, psi_full, stats = plan.eigs_lowest(tol=1e-11, seed=11, comm=comm)
assert stats.converged == 1
y = torch.zeros(info.vec_len, dtype=torch.float64, device=dev)
plan.apply(psi_full, y[info.local_offset:info.local_offset + info.local_len])
comm.allgather(y, info.seg_stride)
res = float((y - e0 * psi_full).norm())
assert res < 1e-8 * abs(e0), res
if rank == 0:
    single = KronPlan(sb)
    e_ref, psi_ref, _ = single.eigs_lowest(tol=1e-11, seed=11)
    psi = torch.zeros(sb.n_states, dtype=torch.float64, device=dev)
    plan.from_striped(psi_full, psi)
    assert abs(e0 - e_ref) <= 1e-10 * abs(e_ref), (e0, e_ref)
    assert abs(abs(float(torch.dot(psi, psi_ref))) - 1.0) < 1e-8
    print(f"native {world}-rank eigensolve ok: E0={e0:.12f} matvecs={stats.n_matvec}")
comm.barrier()
comm.destroy()
