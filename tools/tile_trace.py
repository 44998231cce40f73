"""Run one superblock MatMult on the -DDMRGX_TILE_TRACE build and save the per-workgroup stamps (see tools/tile_trace.sh)."""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
from dmrgx_amd import _capi
_capi.LIB_PATH = os.path.join(ROOT, "tools", sys.argv[3] if len(sys.argv) > 3 else "trace", "libdmrgx_hip.so")
from dmrgx_amd.superblock import KronPlan
from dmrgx_amd.workloads import synthetic_superblock
wl, out = sys.argv[1], sys.argv[2]
sb = synthetic_superblock(wl)
plan = KronPlan(sb, device="cuda:0")
L = _capi.lib()
info = plan.info
x = torch.randn(info.vec_len, dtype=torch.float64, device="cuda")
y = torch.zeros(info.vec_len, dtype=torch.float64, device="cuda")
for _ in range(5):
    plan.apply(x, y)
torch.cuda.synchronize()
cap = 1 << 22
buf = torch.zeros(cap, dtype=torch.int64, device="cuda")
L.dmrgx_debug_tile_trace.argtypes = [C.c_void_p, C.c_size_t]
L.dmrgx_debug_tile_trace_used.restype = C.c_size_t
reps = 3
L.dmrgx_debug_tile_trace(C.c_void_p(buf.data_ptr()), cap)
for _ in range(reps):
    plan.apply(x, y)
torch.cuda.synchronize()
used = L.dmrgx_debug_tile_trace_used()
L.dmrgx_debug_tile_trace(None, 0)
np.save(os.path.join(out, "stamps.npy"), buf[:used].cpu().numpy().reshape(-1, 8))
open(os.path.join(out, "meta.txt"), "w").write(f"{wl} reps {reps} ntiles1 {info.n_tiles_stage1} ntiles2 {info.n_tiles_stage2} flops_alg {info.flops_alg} flops_exec {info.flops_exec}\n")
print("stamps", used // 8)
