"""Worker of test_rccl_hooks_single_rank: the bench's RCCL hooks (collectives.torch_hooks) on a one-rank process group:
checks the zero-copy pointer views and the in-place all_gather_into_tensor / all_reduce call forms on the device."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
load_package()
import torch_collective_hooks as collectives

hooks = collectives.torch_hooks(dist, 0, 1)
full = torch.arange(4096, dtype=torch.float64, device="cuda")
want = full.clone()
st = torch.cuda.current_stream().cuda_stream
assert hooks["allgather"](None, full.data_ptr(), 4096, st) == 0
few = torch.tensor([1.5, -2.0, 3.25], dtype=torch.float64, device="cuda")
assert hooks["allreduce"](None, few.data_ptr(), 3, st) == 0
torch.cuda.synchronize()
assert torch.equal(full, want)
assert few.tolist() == [1.5, -2.0, 3.25]
dist.barrier()
dist.destroy_process_group()
print("rccl hooks ok")
