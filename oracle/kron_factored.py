"""Oracle / CPU baseline: the superblock MatMult in FACTORED, operator-merged form on the host cores.

TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests/): never imported by the product.

SURVEY.md section 8d asks for two CPU statements of reference src/DMRGKron.cpp:1827-1869 beside the GPU number:
  (i)  kron_apply_ref  -- the literal unfactored row loop (oracle/kron_ref.c), what the reference executes;
  (ii) kron_apply_cpu  -- this file: the same operator as per-sector dense GEMMs,
           Y_k = H_L[IL] X_k + X_k H_R[IR]^T + sum_g Abar_g[IL->IL'] (X_k' Bhat_g[IR->IR']^T),
       with the terms that share a right operator merged at set-up (Abar_g = sum_t a_t A_t, the map the reference builds at
       src/DMRGKron.cpp:955-960) and the structural zeros of O (x) 1_2 never multiplied (cell by cell) -- i.e. exactly the
       algorithmic flop count F_alg of SURVEY 8d, executed by numpy's OpenBLAS on all cores.  It is the honest CPU
       counterpart of the HIP plan: same algorithm, CPU hardware.
"""
import numpy as np

OpSm, OpSz, OpSp = -1, 0, +1
CELL_DENSE, CELL_IDENT = 1, 2


def _cells_as_used(op, kind):
    """Cells of a stored operator (Sz or Sp) as used in a term: kind == OpSm reads Sp transposed (src/DMRGBlock.cpp:630-632).
    -> list of (row_sector, r0, c0, nr, nc, kind, scale, array), shift of the operator as used."""
    out = []
    for c in op.cells:
        if kind == OpSm:
            out.append((c.row_sector + op.shift, c.c0, c.r0, c.nc, c.nr, c.kind, c.scale, None if c.array is None else np.ascontiguousarray(c.array.T)))
        else:
            out.append((c.row_sector, c.r0, c.c0, c.nr, c.nc, c.kind, c.scale, c.array))
    return out, (-op.shift if kind == OpSm else op.shift)


class FactoredApplyCPU:
    def __init__(self, sb):
        self.sb = sb
        self.off = sb.block_offsets()
        self.kmap = {b: k for k, b in enumerate(sb.blocks)}
        groups = {}
        for (a, Iop, Isite, Jop, Jsite) in sb.terms:
            if a == 0.0:
                continue
            groups.setdefault((Jop, Jsite), []).append((a, Iop, Isite))
        self.groups = []
        self.flops = 0.0
        for (Jop, Jsite), lst in groups.items():
            rcells, sB = _cells_as_used(sb.right_ops[(OpSp if Jop != OpSz else OpSz, Jsite)], Jop)
            merged = {}
            for (a, Iop, Isite) in lst:
                cells, sA = _cells_as_used(sb.left_ops[(OpSp if Iop != OpSz else OpSz, Isite)], Iop)
                assert sA == -sB
                for (q, r0, c0, nr, nc, kind, scale, arr) in cells:
                    key = (q, r0, c0, nr, nc, kind)
                    if key not in merged:
                        merged[key] = [0.0, None]
                    if kind == CELL_IDENT:
                        merged[key][0] += a * scale
                    else:
                        merged[key][1] = a * arr if merged[key][1] is None else merged[key][1] + a * arr
            lcells = [(q, r0, c0, nr, nc, kind, v[0], v[1]) for (q, r0, c0, nr, nc, kind), v in merged.items()]
            self.groups.append((-sB, sB, lcells, rcells))
        self.hl, _ = _cells_as_used(sb.h_left, OpSz)
        self.hr, _ = _cells_as_used(sb.h_right, OpSz)
        # algorithmic flops (SURVEY 8d F_alg), counted once from the cells
        for k, (il, ir) in enumerate(sb.blocks):
            nl, nr_ = sb.left_sizes[il], sb.right_sizes[ir]
            for (q, r0, c0, nr, nc, kind, scale, arr) in self.hl:
                if q == il:
                    self.flops += 2.0 * nr * nc * nr_ if kind == CELL_DENSE else 2.0 * nr * nr_
            for (q, r0, c0, nr, nc, kind, scale, arr) in self.hr:
                if q == ir:
                    self.flops += 2.0 * nl * nr * nc if kind == CELL_DENSE else 2.0 * nl * nr
            for (sA, sB, lcells, rcells) in self.groups:
                ks = self.kmap.get((il + sA, ir + sB))
                if ks is None:
                    continue
                nls = sb.left_sizes[il + sA]
                for (q, r0, c0, nr, nc, kind, scale, arr) in rcells:
                    if q == ir:
                        self.flops += 2.0 * nls * nr * nc if kind == CELL_DENSE else 2.0 * nls * nr
                for (q, r0, c0, nr, nc, kind, scale, arr) in lcells:
                    if q == il:
                        self.flops += 2.0 * nr * nc * nr_ if kind == CELL_DENSE else 2.0 * nr * nr_

    def apply(self, x):
        sb, off = self.sb, self.off
        X = [x[off[k]:off[k + 1]].reshape(sb.left_sizes[il], sb.right_sizes[ir]) for k, (il, ir) in enumerate(sb.blocks)]
        y = np.empty_like(x)
        for k, (il, ir) in enumerate(sb.blocks):
            nl, nr_ = sb.left_sizes[il], sb.right_sizes[ir]
            Y = np.zeros((nl, nr_))
            for (q, r0, c0, nr, nc, kind, scale, arr) in self.hl:                  # H_L (x) 1
                if q == il:
                    Y[r0:r0 + nr] += (arr @ X[k][c0:c0 + nc]) if kind == CELL_DENSE else scale * X[k][c0:c0 + nr]
            for (q, r0, c0, nr, nc, kind, scale, arr) in self.hr:                  # 1 (x) H_R: Y[:, r] += X[:, c] H_R[r, c]
                if q == ir:
                    if kind == CELL_DENSE:
                        Y[:, r0:r0 + nr] += X[k][:, c0:c0 + nc] @ arr.T
                    else:
                        Y[:, r0:r0 + nr] += scale * X[k][:, c0:c0 + nr]
            for (sA, sB, lcells, rcells) in self.groups:
                ks = self.kmap.get((il + sA, ir + sB))
                if ks is None:
                    continue
                T = np.zeros((sb.left_sizes[il + sA], nr_))                        # T = X_k' Bhat^T
                for (q, r0, c0, nr, nc, kind, scale, arr) in rcells:
                    if q != ir:
                        continue
                    if kind == CELL_DENSE:
                        T[:, r0:r0 + nr] += X[ks][:, c0:c0 + nc] @ arr.T
                    else:
                        T[:, r0:r0 + nr] += scale * X[ks][:, c0:c0 + nr]
                for (q, r0, c0, nr, nc, kind, scale, arr) in lcells:
                    if q != il:
                        continue
                    if kind == CELL_DENSE:
                        Y[r0:r0 + nr] += arr @ T[c0:c0 + nc]
                    else:
                        Y[r0:r0 + nr] += scale * T[c0:c0 + nr]
            y[off[k]:off[k + 1]] = Y.ravel()
        return y
