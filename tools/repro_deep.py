import os, sys, math
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
load_package()
from dmrgx_amd.superblock import KronPlan
from dmrgx_amd.workloads import synthetic_superblock
from helpers import oracle_shell_from_superblock
from oracle.kron_c import ShellApplyC
def binom_profile(n):
    return {n / 2 - k: math.comb(n, k) for k in range(n + 1)}
for (nl, nr) in ((8, 6), (9, 7), (9, 9), (10, 8), (7, 5)):
    sb = synthetic_superblock("cfg4", kept=(binom_profile(nl), binom_profile(nr)), seed=5)
    plan = KronPlan(sb, device="cuda:0")
    rng = np.random.default_rng(0)
    x = rng.standard_normal(sb.n_states)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros_like(xd)
    plan.apply(xd, yd); torch.cuda.synchronize()
    y_ref = ShellApplyC(oracle_shell_from_superblock(sb)).apply(x)
    y = yd.cpu().numpy()
    err = np.abs(y - y_ref).max() / np.abs(y_ref).max()
    print(f"kept {nl}|{nr} sites: tiles {plan.info.n_tiles_stage1}+{plan.info.n_tiles_stage2} n_states {sb.n_states} sizes L {sb.left_sizes} rel err {err:.2e}", flush=True)
    if err > 1e-12:
        off = sb.block_offsets()
        for k, (il, ir) in enumerate(sb.blocks):
            e = np.abs(y[off[k]:off[k+1]] - y_ref[off[k]:off[k+1]]).reshape(sb.left_sizes[il], sb.right_sizes[ir])
            if e.max() > 1e-10:
                bad = np.argwhere(e > 1e-10)
                print(f"  block {k} ({sb.left_sizes[il]} x {sb.right_sizes[ir]}): max {e.max():.2e}, bad rows {bad[:,0].min()}..{bad[:,0].max()} cols {bad[:,1].min()}..{bad[:,1].max()} count {len(bad)}")
    plan.destroy()
