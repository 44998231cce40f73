"""Collective hooks of the eigensolver for world_size > 1, over torch.distributed (backend "nccl" == RCCL).

The C ABI asks for two stream-ordered callbacks (include/dmrgx.h, dmrgx_eigs_opts): an in-place all-gather of
the Krylov vector's rank segments and a sum all-reduce of a few doubles.  They replace the reference's
VecScatter-to-all inside every MatMult (src/DMRGKron.cpp:1833-1834) and the MPI_Allreduce behind SLEPc's
VecDot/VecNorm.  Raw device pointers from the library are wrapped zero-copy as torch tensors.
"""
import torch


class _Raw:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def _view(ptr, n, device):
    return torch.as_tensor(_Raw(ptr, n), device=device)


def torch_hooks(dist, rank, world):
    device = torch.device("cuda", torch.cuda.current_device())
    views = {}          # (pointer, length) -> tensor view: the eigensolver calls back with the same few buffers every step

    def view(ptr, n):
        key = (int(ptr), int(n))
        t = views.get(key)
        if t is None:
            t = views[key] = _view(ptr, n, device)
        return t

    def allgather(user, full_ptr, seg_stride, stream):
        full = view(full_ptr, seg_stride * world)
        # out-of-place send buffer: a copy of this rank's segment (a few MB over HBM) instead of relying on aliasing rules
        # of the in-place form
        dist.all_gather_into_tensor(full, full[rank * seg_stride:(rank + 1) * seg_stride].clone())
        return 0

    def allreduce(user, buf_ptr, count, stream):
        dist.all_reduce(view(buf_ptr, count))
        return 0

    return {"allgather": allgather, "allreduce": allreduce}


def host_staged_hooks(dist, rank, world):
    """The same two hooks with the payload staged through host memory (backend gloo): for rehearsing the N > 1 control
    flow with several ranks on ONE GPU (tests); never used for measurements."""
    device = torch.device("cuda", torch.cuda.current_device())

    def allgather(user, full_ptr, seg_stride, stream):
        full = _view(full_ptr, seg_stride * world, device)
        mine = full[rank * seg_stride:(rank + 1) * seg_stride].cpu()
        out = torch.empty(seg_stride * world, dtype=torch.float64)
        dist.all_gather_into_tensor(out, mine)
        full.copy_(out)
        return 0

    def allreduce(user, buf_ptr, count, stream):
        v = _view(buf_ptr, count, device)
        h = v.cpu()
        dist.all_reduce(h)
        v.copy_(h)
        return 0

    return {"allgather": allgather, "allreduce": allreduce}
