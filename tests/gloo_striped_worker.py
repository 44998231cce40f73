"""Worker of test_world_size_2_striped_apply_over_gloo: rank-major striped superblock apply over gloo (CPU)."""
import ctypes as C
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
pkg = load_package()
from dmrgx_amd import workloads as wl
lib = pkg._capi.lib()

sb = wl.synthetic_superblock("cfg2", m=40, Ly=3, seed=21)
off = sb.block_offsets()
x = np.random.default_rng(3).standard_normal(sb.n_states)
y_full = wl.apply_factored_numpy(sb, x)


def bounds(n, r, k):
    a, b = C.c_int32(), C.c_int32()
    assert lib.dmrgx_stripe_bounds_of_block(n, world, r, k, C.byref(a), C.byref(b)) == 0
    return a.value, b.value


def segment(vec, r):
    """rank r's segment of a full vector: for every KronBlock the columns of its stripe, row-major."""
    parts = []
    for k, (il, ir) in enumerate(sb.blocks):
        c0, c1 = bounds(sb.right_sizes[ir], r, k)
        parts.append(vec[off[k]:off[k + 1]].reshape(sb.left_sizes[il], sb.right_sizes[ir])[:, c0:c1].ravel())
    return np.concatenate(parts)


seg_len = [segment(x, r).size for r in range(world)]
stride = ((max(seg_len) + 63) // 64) * 64
# each rank applies H to the full x but keeps only its stripe (the device plan computes just that stripe)
mine = np.zeros(stride)
mine[:seg_len[rank]] = segment(y_full, rank)
gathered = torch.zeros(world * stride, dtype=torch.float64)
dist.all_gather_into_tensor(gathered, torch.from_numpy(mine))
g = gathered.numpy()
# reassemble the reference layout from the rank-major segments
y = np.zeros(sb.n_states)
for r in range(world):
    pos = r * stride
    for k, (il, ir) in enumerate(sb.blocks):
        c0, c1 = bounds(sb.right_sizes[ir], r, k)
        n = sb.left_sizes[il] * (c1 - c0)
        y[off[k]:off[k + 1]].reshape(sb.left_sizes[il], sb.right_sizes[ir])[:, c0:c1] = g[pos:pos + n].reshape(sb.left_sizes[il], c1 - c0)
        pos += n
assert np.array_equal(y, y_full)
# the fused reductions of a Lanczos step: local partial dots summed by all-reduce equal the global dots
t = torch.tensor([float(np.dot(segment(x, rank), segment(y_full, rank))), float(np.dot(segment(x, rank), segment(x, rank)))], dtype=torch.float64)
dist.all_reduce(t)
assert abs(t[0].item() - np.dot(x, y_full)) < 1e-9 * abs(np.dot(x, y_full)) and abs(t[1].item() - np.dot(x, x)) < 1e-9 * np.dot(x, x)
if rank == 0:
    print("striped apply ok")
dist.destroy_process_group()
