"""Worker of test_two_rank_eigensolve_on_one_gpu: two ranks share cuda:0, each owns one right-index stripe of the
superblock; the eigensolver's collective hooks are staged through gloo on the host (RCCL needs one GPU per rank,
this checks everything else of the N>1 path: striped plans, hooks, fused reductions, all-gather of the Krylov vector)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package

rank, world = int(sys.argv[1]), int(sys.argv[2])
dist.init_process_group("gloo", rank=rank, world_size=world)
load_package()
from dmrgx_amd.superblock import KronPlan
from dmrgx_amd.workloads import synthetic_superblock
from torch_collective_hooks import _view

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
sb = synthetic_superblock("cfg2", m=64, Ly=3, seed=5)
plan = KronPlan(sb, world_size=world, rank=rank)
info = plan.info
calls = {"ag": 0, "ar": 0}


def allgather(user, full_ptr, seg_stride, stream):
    full = _view(full_ptr, seg_stride * world, dev)
    mine = full[rank * seg_stride:(rank + 1) * seg_stride].cpu()
    out = torch.empty(seg_stride * world, dtype=torch.float64)
    dist.all_gather_into_tensor(out, mine)
    full.copy_(out)
    calls["ag"] += 1
    return 0


def allreduce(user, buf_ptr, count, stream):
    v = _view(buf_ptr, count, dev)
    h = v.cpu()
    dist.all_reduce(h)
    v.copy_(h)
    calls["ar"] += 1
    return 0


e0, psi_full, stats = plan.eigs_lowest(tol=1e-11, seed=11, allgather=allgather, allreduce=allreduce)
assert stats.converged == 1
# residual with the distributed apply: y stripes -> all-gather -> compare
y = torch.zeros(info.vec_len, dtype=torch.float64, device=dev)
plan.apply(psi_full, y[info.local_offset:info.local_offset + info.local_len])
allgather(None, y.data_ptr(), info.seg_stride, None)
res = float((y - e0 * psi_full).norm())
assert res < 1e-8 * abs(e0), res
assert stats.n_matvec > 0 and calls["ag"] >= stats.n_matvec and calls["ar"] <= 2 * stats.n_matvec + 8, (calls, stats.n_matvec)
if rank == 0:
    single = KronPlan(sb)
    e_ref, psi_ref, _ = single.eigs_lowest(tol=1e-11, seed=11)
    psi = torch.zeros(sb.n_states, dtype=torch.float64, device=dev)
    plan.from_striped(psi_full, psi)
    overlap = abs(float(torch.dot(psi, psi_ref)))
    assert abs(e0 - e_ref) <= 1e-10 * abs(e_ref), (e0, e_ref)
    assert abs(overlap - 1.0) < 1e-8, overlap
    print(f"two-rank eigensolve ok: E0={e0:.12f} matvecs={stats.n_matvec} allgathers={calls['ag']} allreduces={calls['ar']}")
dist.barrier()
dist.destroy_process_group()
