"""Test infrastructure: bridges between the product's Superblock description and the CPU oracle."""
import numpy as np
import scipy.sparse as sp

from oracle.block import Block
from oracle.hamiltonian import Term
from oracle.kron import KronBlocks, ShellCtx
from oracle.qn import QuantumNumbers

CELL_DENSE, CELL_IDENT = 1, 2


def operator_to_csr(op, sizes):
    """SectorOperator (cells) -> scipy CSR of the whole block basis."""
    off = np.concatenate([[0], np.cumsum(sizes)])
    n = int(off[-1])
    R, C, V = [], [], []
    for c in op.cells:
        r0, c0 = off[c.row_sector] + c.r0, off[c.row_sector + op.shift] + c.c0
        if c.kind == CELL_DENSE:
            ii, jj = np.meshgrid(np.arange(c.nr), np.arange(c.nc), indexing="ij")
            R.append((r0 + ii).ravel()); C.append((c0 + jj).ravel()); V.append(np.asarray(c.array).ravel())
        else:
            R.append(r0 + np.arange(c.nr)); C.append(c0 + np.arange(c.nr)); V.append(np.full(c.nr, c.scale))
    if not R:
        return sp.csr_matrix((n, n))
    m = sp.coo_matrix((np.concatenate(V), (np.concatenate(R), np.concatenate(C))), shape=(n, n)).tocsr()
    m.sort_indices()
    return m


def oracle_blocks_from_superblock(sb):
    """Oracle Block objects (CSR Sz(i), Sp(i), H + Magnetization) and the un-reflected reference Term list."""
    def mk(nsites, qn, sizes, ops, h):
        b = Block.with_sectors(nsites, qn, sizes)
        for (op, site), o in ops.items():
            (b.SzData if op == 0 else b.SpData)[site] = operator_to_csr(o, sizes)
        b.H = operator_to_csr(h, sizes)
        return b
    L = mk(sb.n_left_sites, sb.left_qn, sb.left_sizes, sb.left_ops, sb.h_left)
    R = mk(sb.n_right_sites, sb.right_qn, sb.right_sizes, sb.right_ops, sb.h_right)
    nout = sb.n_left_sites + sb.n_right_sites
    # the reference reflects right sites itself (src/DMRGKron.cpp:805-807): hand it global indices
    terms = [Term(a, Iop, Isite, Jop, nout - 1 - Jsite) for (a, Iop, Isite, Jop, Jsite) in sb.terms]
    return L, R, terms


def oracle_shell_from_superblock(sb, target=0.0):
    L, R, terms = oracle_blocks_from_superblock(sb)
    kb = KronBlocks(L, R, (target,))
    assert [(t[1], t[2]) for t in kb.kb] == list(sb.blocks), "KronBlock order differs from the reference's nested loop"
    return ShellCtx(kb, terms)


# ---- independent exact diagonalisation of the lattice model (correlator known answers) -------------------------------
def lattice_ground_state(ham, spin="1/2"):
    """Dense ED of the whole lattice in the site basis (site 0 = most significant factor), from the model's own term
    list.  Returns (E0, psi, site_op) where site_op(op, i) is the d^N x d^N matrix of a single-site operator
    (single-site matrices of src/DMRGBlock.cpp:1131-1215 for spin 1/2 and spin 1)."""
    import scipy.sparse as sp
    from oracle.qn import OpSm, OpSz, OpSp
    N = ham.NumSites()
    if spin == "1":
        sz = sp.csr_matrix(np.diag([1.0, 0.0, -1.0]))
        spl = sp.csr_matrix(np.sqrt(2.0) * np.array([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [0.0, 0.0, 0.0]]))
    else:
        sz = sp.csr_matrix(np.array([[0.5, 0.0], [0.0, -0.5]]))
        spl = sp.csr_matrix(np.array([[0.0, 1.0], [0.0, 0.0]]))
    d = sz.shape[0]
    single = {OpSz: sz, OpSp: spl, OpSm: spl.T.tocsr()}
    cache = {}

    def site_op(op, i):
        if (op, i) not in cache:
            m = sp.identity(1, format="csr")
            for s in range(N):
                m = sp.kron(m, single[op] if s == i else sp.identity(d, format="csr"), format="csr")
            cache[(op, i)] = m
        return cache[(op, i)]

    H = sp.csr_matrix((d ** N, d ** N))
    for t in ham.H(N):
        H = H + t.a * (site_op(t.Iop, t.Isite) @ site_op(t.Jop, t.Jsite))
    w, v = np.linalg.eigh(H.toarray())
    return float(w[0]), v[:, 0].copy(), site_op


def parse_desc2(desc2):
    """'< Sz_{3} Sp_{4} >' -> [(OpSz, 3), (OpSp, 4)] (the engine / reference description string of a correlator)."""
    import re
    from oracle.qn import OpSm, OpSz, OpSp
    kinds = {"Sz": OpSz, "Sp": OpSp, "Sm": OpSm}
    return [(kinds[k], int(i)) for k, i in re.findall(r"(S[zpm])_\{(\d+)\}", desc2)]
