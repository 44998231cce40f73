"""Synthetic superblock workloads with the sector structure of a mid-sweep DMRG step.

Pure numpy, no device code and no oracle import: the same `Superblock` description feeds the HIP path
(superblock.KronPlan), the CPU oracle (tests, cpu_baseline) and the flop/byte accounting.

Sector profile of a kept block with m states: n(q) = floor(m*w_q/sum w), w_q = exp(-q^2/(2 sigma^2)),
integer q in [-ceil(5 sigma), ceil(5 sigma)], sigma = 1.8, remainder added to q = 0, empty sectors dropped.
An enlarged block (kept block (x) one spin-1/2 site) has sectors at q_e = q +- 1/2 with
n_enl(q_e) = n(q_e+1/2) + n(q_e-1/2), ordered inside a sector as [old(q_e+1/2) (x) down ; old(q_e-1/2) (x) up],
which is the merged-KronBlock order of the reference (src/DMRGKron.cpp:561-574 with the stable sort of
include/DMRGKron.hpp:157).  Operators of old sites are O (x) 1_2 (two dense cells per sector block), the new
site's operators are 1 (x) s (scaled-identity cells), H_L/H_R are dense symmetric per sector.
"""
import math
from dataclasses import dataclass, field

import numpy as np

OpSm, OpSz, OpSp = -1, 0, +1
CELL_DENSE, CELL_IDENT = 1, 2


@dataclass
class OpCell:
    row_sector: int
    r0: int
    c0: int
    nr: int
    nc: int
    kind: int = CELL_DENSE
    scale: float = 0.0
    array: np.ndarray = None   # nr x nc, C-contiguous f64 (DENSE only)


@dataclass
class SectorOperator:
    shift: int                 # column sector = row sector + shift (Op_t value)
    cells: list = field(default_factory=list)

    def nnz(self):
        return sum(c.nr * c.nc if c.kind == CELL_DENSE else c.nr for c in self.cells)


@dataclass
class Superblock:
    """Everything KronSumConstructShell sees (reference src/DMRGKron.cpp:1871-1917), in sector-cell form."""
    name: str
    left_sizes: list
    right_sizes: list
    left_qn: list              # Sz of each sector, descending
    right_qn: list
    blocks: list               # [(IL, IR)] target-sector KronBlocks in the reference's order
    left_ops: dict             # (OpSz|OpSp, site) -> SectorOperator   (Sm(i) is Sp(i) transposed, never stored)
    right_ops: dict
    h_left: SectorOperator
    h_right: SectorOperator
    terms: list                # [(a, Iop, Isite, Jop, Jsite)] with block-local (already reflected) site indices
    n_left_sites: int
    n_right_sites: int

    @property
    def n_states(self):
        return sum(self.left_sizes[il] * self.right_sizes[ir] for il, ir in self.blocks)

    def block_offsets(self):
        off = [0]
        for il, ir in self.blocks:
            off.append(off[-1] + self.left_sizes[il] * self.right_sizes[ir])
        return off


def kept_profile(m, sigma=1.8):
    Q = int(math.ceil(5 * sigma))
    qs = list(range(Q, -Q - 1, -1))
    w = np.array([math.exp(-q * q / (2 * sigma * sigma)) for q in qs])
    n = np.floor(m * w / w.sum()).astype(int)
    n[qs.index(0)] += m - int(n.sum())
    return {q: int(c) for q, c in zip(qs, n) if c > 0}


def enlarged_sectors(kept):
    """kept: {Sz of a kept-block sector: size} (Sz integer or half-integer).
    -> (qn list desc, sizes, sub = [(size_down_part, size_up_part)]) of kept (x) one spin-1/2 site."""
    kept2 = {int(round(2 * q)): n for q, n in kept.items()}          # keyed by 2*Sz
    two_q = sorted({t + 1 for t in kept2} | {t - 1 for t in kept2}, reverse=True)
    qn, sizes, sub = [], [], []
    for tq in two_q:
        dn = kept2.get(tq + 1, 0)   # old sector q_e + 1/2, new site down
        up = kept2.get(tq - 1, 0)   # old sector q_e - 1/2, new site up
        if dn + up == 0:
            continue
        qn.append(tq / 2.0)
        sizes.append(dn + up)
        sub.append((dn, up))
    return qn, sizes, sub


def _old_site_op(rng, shift, sizes, sub):
    """O (x) 1_2 for an operator O of the kept block (random dense sector blocks)."""
    op = SectorOperator(shift)
    for q in range(len(sizes)):
        qc = q + shift
        if not (0 <= qc < len(sizes)):
            continue
        (rd, ru), (cd, cu) = sub[q], sub[qc]

        def rnd(a, b):
            mat = rng.standard_normal((a, b))
            if shift == OpSz:      # Sz(i) of a real block is symmetric; keeps the synthetic H_sb symmetric
                mat = (mat + mat.T) * 0.5
            return np.ascontiguousarray(mat)
        if rd and cd:
            op.cells.append(OpCell(q, 0, 0, rd, cd, CELL_DENSE, 0.0, rnd(rd, cd)))
        if ru and cu:
            op.cells.append(OpCell(q, rd, cd, ru, cu, CELL_DENSE, 0.0, rnd(ru, cu)))
    return op


def _new_site_op(shift, sizes, sub):
    """1 (x) s for the added spin-1/2 site: Sz = diag(-1/2 | +1/2), Sp = |up><down| (src/DMRGBlock.cpp:1131-1136,1193-1195)."""
    op = SectorOperator(shift)
    for q in range(len(sizes)):
        rd, ru = sub[q]
        if shift == OpSz:
            if rd:
                op.cells.append(OpCell(q, 0, 0, rd, rd, CELL_IDENT, -0.5))
            if ru:
                op.cells.append(OpCell(q, rd, rd, ru, ru, CELL_IDENT, +0.5))
        else:  # Sp: row (old p, up) in sector q  <-  column (old p, down) in sector q+1
            qc = q + 1
            if qc < len(sizes) and ru and sub[qc][0]:
                assert sub[qc][0] == ru
                op.cells.append(OpCell(q, rd, 0, ru, ru, CELL_IDENT, 1.0))
    return op


def _sym_block_op(rng, sizes):
    op = SectorOperator(0)
    for q, n in enumerate(sizes):
        a = rng.standard_normal((n, n))
        op.cells.append(OpCell(q, 0, 0, n, n, CELL_DENSE, 0.0, np.ascontiguousarray((a + a.T) * 0.5)))
    return op


# Kept-block sector tables of a REAL mid-sweep step of BASELINE configs[3] (J1-J2 20x8 cylinder, J2 = 0.5, m = 2048), dumped by
# the engine into KronStats.json (second sweep, GlobIdx 228: system block of 80 sites, environment block of 78 sites;
# profiles/r02_cfg4_sweep_kronstats.json).  The real distribution is narrower than SURVEY 8d's sigma = 1.8 estimate
# (central sectors of 578 / 459 states instead of 462 / 388), which makes the real superblock 1.7x heavier per MatMult.
REAL_PROFILES = {
    "cfg4real": dict(left={4: 4, 3: 52, 2: 220, 1: 459, 0: 578, -1: 459, -2: 220, -3: 52, -4: 4},
                     right={4: 6, 3: 58, 2: 220, 1: 453, 0: 575, -1: 453, -2: 220, -3: 57, -4: 6}),
}

CONFIGS = {
    # name: (m, Ly, J1, Jz1, J2, Jz2, seed)   -- BASELINE.json configs[0..4]; seeds per SURVEY 8d
    "cfg1": dict(m=64, Ly=1, J1=0.5, Jz1=1.0, J2=0.0, Jz2=0.0, seed=20261, desc="1D Heisenberg chain 16x1, m=64"),
    "cfg2": dict(m=512, Ly=4, J1=1.0, Jz1=1.0, J2=0.5, Jz2=0.5, seed=20262, desc="J1-J2 8x4 cylinder, J2=0.5, m=512"),
    "cfg3": dict(m=1024, Ly=6, J1=0.5, Jz1=1.0, J2=0.0, Jz2=0.0, seed=20263, desc="Heisenberg 16x6 cylinder, m=1024"),
    "cfg4": dict(m=2048, Ly=8, J1=1.0, Jz1=1.0, J2=0.5, Jz2=0.5, seed=20264, desc="J1-J2 20x8 cylinder, J2=0.5, m=2048"),
    "cfg4real": dict(m=2048, Ly=8, J1=1.0, Jz1=1.0, J2=0.5, Jz2=0.5, seed=20264,
                     desc="J1-J2 20x8 cylinder, J2=0.5, m=2048, sector tables of a real mid-sweep step (80|78 sites) of the engine's second sweep"),
    "cfg5": dict(m=4096, Ly=8, J1=1.0, Jz1=0.0, J2=0.0, Jz2=0.0, seed=20265, desc="XY 32x8 cylinder (no NNN: reference quirk), m=4096"),
}


def column_cut_terms(Ly, J1, Jz1, J2, Jz2):
    """Inter-block terms at a column-aligned cut of a width-Ly cylinder (open x, periodic y).

    Left boundary site i and right boundary site j (block-local index along the column, 0..Ly-1; the new site
    of each enlarged block is index Ly-1).  NN bond i==j; NNN bonds |i-j|==1 (mod Ly).  Each bond contributes
    S+S-, S-S+ (coefficient J) and SzSz (Jz) exactly as src/Hamiltonians.cpp:93-95,110-112; NNN only if both
    J2 and Jz2 are non-zero (src/Hamiltonians.cpp:101); for Ly == 2 periodic-y doubles the diagonal bond.
    """
    bonds = [(i, i, J1, Jz1) for i in range(Ly)]
    if J2 != 0.0 and Jz2 != 0.0 and Ly > 1:
        for i in range(Ly):
            for dj in (+1, -1):
                j = i + dj
                if Ly > 2:
                    j %= Ly
                elif not (0 <= j < Ly):
                    continue
                bonds.append((i, j, J2, Jz2))
    terms = []
    for (i, j, J, Jz) in bonds:
        if J != 0.0:
            terms.append((J, OpSp, i, OpSm, j))
            terms.append((J, OpSm, i, OpSp, j))
        if Jz != 0.0:
            terms.append((Jz, OpSz, i, OpSz, j))
    return terms


def scaled_real_profile(name, factor):
    """The kept-sector tables of a real sweep step (REAL_PROFILES), every sector divided by `factor` (at least one state): the
    L != R structure of the bench workload at a size the CPU oracle's row loop can check."""
    p = REAL_PROFILES[name]
    return ({q: max(1, int(round(n / factor))) for q, n in p["left"].items()}, {q: max(1, int(round(n / factor))) for q, n in p["right"].items()})


def synthetic_superblock(name="cfg2", m=None, Ly=None, seed=None, sigma=1.8, kept=None, **couplings):
    """Mid-chain, column-aligned superblock of BASELINE config `name` (or custom m/Ly/couplings).
    kept = (left, right): explicit kept-sector tables {Sz: states} of the two blocks (overrides m / the real profile)."""
    cfg = dict(CONFIGS.get(name, CONFIGS["cfg2"]))
    if m is not None:
        cfg["m"] = m
    if Ly is not None:
        cfg["Ly"] = Ly
    if seed is not None:
        cfg["seed"] = seed
    cfg.update(couplings)
    rng = np.random.default_rng(cfg["seed"])
    if kept is not None:
        kept_l, kept_r = kept
    elif name in REAL_PROFILES and m is None:
        kept_l, kept_r = REAL_PROFILES[name]["left"], REAL_PROFILES[name]["right"]
    else:
        kept_l = kept_r = kept_profile(cfg["m"], sigma)
    qn, sizes, sub = enlarged_sectors(kept_l)
    rqn, rsizes, rsub = (qn, sizes, sub) if kept_r is kept_l else enlarged_sectors(kept_r)
    Ly = cfg["Ly"]
    terms = column_cut_terms(Ly, cfg["J1"], cfg["Jz1"], cfg["J2"], cfg["Jz2"])

    def side_ops(used, sizes, sub):
        ops = {}
        for (op, site) in sorted(used):
            base = OpSp if op in (OpSp, OpSm) else OpSz
            if (base, site) in ops:
                continue
            if site == Ly - 1:
                ops[(base, site)] = _new_site_op(base, sizes, sub)
            else:
                ops[(base, site)] = _old_site_op(rng, base, sizes, sub)
        return ops

    left_ops = side_ops({(t[1], t[2]) for t in terms}, sizes, sub)
    right_ops = side_ops({(t[3], t[4]) for t in terms}, rsizes, rsub)
    h_left, h_right = _sym_block_op(rng, sizes), _sym_block_op(rng, rsizes)
    # target sector Sz_total = 0: q_L + q_R == 0, nested IL-then-IR order (include/DMRGKron.hpp:160-171)
    blocks = [(il, ir) for il in range(len(qn)) for ir in range(len(rqn)) if qn[il] + rqn[ir] == 0.0]
    return Superblock(name=name, left_sizes=list(sizes), right_sizes=list(rsizes), left_qn=list(qn), right_qn=list(rqn),
                      blocks=blocks, left_ops=left_ops, right_ops=right_ops, h_left=h_left, h_right=h_right,
                      terms=terms, n_left_sites=Ly, n_right_sites=Ly)


def operator_to_dense_blocks(op, sizes):
    """{row sector: dense n_q x n_{q+shift} block} (zeros where no cell)."""
    out = {}
    for c in op.cells:
        qc = c.row_sector + op.shift
        blk = out.setdefault(c.row_sector, np.zeros((sizes[c.row_sector], sizes[qc])))
        if c.kind == CELL_DENSE:
            blk[c.r0:c.r0 + c.nr, c.c0:c.c0 + c.nc] += c.array
        else:
            blk[c.r0 + np.arange(c.nr), c.c0 + np.arange(c.nr)] += c.scale
    return out


def apply_factored_numpy(sb, x):
    """y = H x in factored per-KronBlock form with numpy GEMMs (independent of both the HIP tables and the
    reference's row loop): Y_k += a * A[IL->IL'] X_k' B[IR->IR']^T."""
    off = sb.block_offsets()
    kmap = {b: k for k, b in enumerate(sb.blocks)}
    y = np.zeros_like(x)
    dl = {key: operator_to_dense_blocks(op, sb.left_sizes) for key, op in sb.left_ops.items()}
    dr = {key: operator_to_dense_blocks(op, sb.right_sizes) for key, op in sb.right_ops.items()}
    hl = operator_to_dense_blocks(sb.h_left, sb.left_sizes)
    hr = operator_to_dense_blocks(sb.h_right, sb.right_sizes)

    def blk(dense, op, q, nsec):
        """block (q -> q+op) of operator `op` of a site; Sm block = transpose of the Sp block (q+op -> q)."""
        if op == OpSm:
            b = dense.get(q - 1)
            return None if b is None else b.T
        return dense.get(q)

    X = [x[off[k]:off[k + 1]].reshape(sb.left_sizes[il], sb.right_sizes[ir]) for k, (il, ir) in enumerate(sb.blocks)]
    for k, (il, ir) in enumerate(sb.blocks):
        Y = np.zeros((sb.left_sizes[il], sb.right_sizes[ir]))
        if il in hl:
            Y += hl[il] @ X[k]
        if ir in hr:
            Y += X[k] @ hr[ir].T
        for (a, Iop, Isite, Jop, Jsite) in sb.terms:
            sA = Iop
            ks = kmap.get((il + sA, ir - sA))
            if ks is None:
                continue
            A = blk(dl[(OpSp if Iop != OpSz else OpSz, Isite)], Iop, il, len(sb.left_sizes))
            Bm = blk(dr[(OpSp if Jop != OpSz else OpSz, Jsite)], Jop, ir, len(sb.right_sizes))
            if A is None or Bm is None:
                continue
            Y += a * (A @ X[ks] @ Bm.T)
        y[off[k]:off[k + 1]] = Y.ravel()
    return y
