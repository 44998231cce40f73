"""CPU suite: the oracle against every known-answer table the reference's own tests hold, against the single-site
operator definitions, and against independent exact-diagonalisation energies."""
import json
import os

import numpy as np
import pytest

from oracle.block import Block, csr_from_rows
from oracle.dmrg import DMRGOracle, GetTruncation, lowest_eigenpair
from oracle.hamiltonian import J1J2XXZModel_SquareLattice, Term
from oracle.kron import KronBlocks, KronEye_Explicit, KronSumConstruct_explicit, ShellCtx
from oracle.kron_c import ShellApplyC
from oracle.qn import QuantumNumbers, OracleError, OpSm, OpSz, OpSp, PETSC_ERR_ARG_OUTOFRANGE

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _block_from_fixture(d):
    b = Block.with_sectors(d["nsites"], d["qn_list"], d["qn_size"])
    n = b.NumStates()
    for opname, dst in (("Sz", b.SzData), ("Sp", b.SpData)):
        for site, rows in d[opname].items():
            # SetRow stores value == column index (tests/UnitTests_Misc.cpp:17)
            dst[int(site)] = csr_from_rows(n, {int(r): [(c, float(c)) for c in cols] for r, cols in rows.items()})
    b.H = csr_from_rows(n, {})
    return b


@pytest.mark.parametrize("literal", [True, False])
def test_kron01_known_answer_table(literal):
    """tests/UnitTests_DMRGKron.cpp:39-252: every row of the 10 output operators of KronEye_Explicit."""
    g = json.load(open(os.path.join(GOLD, "testkron01.json")))
    L, R = _block_from_fixture(g["left"]), _block_from_fixture(g["right"])
    out = KronEye_Explicit(L, R, [], literal=literal)
    assert out.NumStates() == 12 and out.NumSites() == 5
    out.CheckOperatorBlocks()
    nchecked = 0
    for opname, arr in (("Sz", out.SzData), ("Sp", out.SpData)):
        for site, rows in g["expected"][opname].items():
            m = arr[int(site)]
            for row, exp in rows.items():
                a, b = m.indptr[int(row)], m.indptr[int(row) + 1]
                assert list(m.indices[a:b]) == exp["cols"], (opname, site, row)
                assert list(m.data[a:b]) == exp["vals"], (opname, site, row)
                nchecked += 1
    assert nchecked == 120


def test_kron01_sector_pair_ordering():
    """Merged sectors of TestKron01: (+1/2,-1/2) x (+1,0,-1) -> qn {1.5,.5,-.5,-1.5}, sizes {2,5,4,1} (src/DMRGKron.cpp:561-574)."""
    g = json.load(open(os.path.join(GOLD, "testkron01.json")))
    L, R = _block_from_fixture(g["left"]), _block_from_fixture(g["right"])
    kb = KronBlocks(L, R, ())
    assert [(t[1], t[2]) for t in kb.kb] == [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (1, 2)]
    out = KronEye_Explicit(L, R, [])
    assert out.Magnetization.qn_list == [1.5, 0.5, -0.5, -1.5] and out.Magnetization.qn_size == [2, 5, 4, 1]


def test_block_fixture_valid_and_planted_error():
    """tests/UnitTests_Misc.cpp:82-136 patterns pass; the planted row of tests/UnitTests_DMRGBlock.cpp:112 raises
    PETSC_ERR_ARG_OUTOFRANGE exactly like the reference's own test expects (:120)."""
    f = json.load(open(os.path.join(GOLD, "block_fixture.json")))
    b = Block.with_sectors(f["nsites"], f["qn_list"], f["qn_size"])
    n = b.NumStates()
    mk = lambda rows: csr_from_rows(n, {int(r): [(c, float(c)) for c in cols] for r, cols in rows.items()})
    b.SzData = [mk(f["valid"]["SetSz0"]), mk(f["valid"]["SetSz1"])]
    b.SpData = [mk(f["valid"]["SetSp0"]), mk(f["valid"]["SetSp1"])]
    b.CheckOperatorBlocks()
    b.SzData[1] = mk(f["planted"]["Sz1"])
    with pytest.raises(OracleError) as ei:
        b.MatOpCheckOperatorBlocks(OpSz, 1)
    assert ei.value.code == f["planted_bad"]["expect_code"] == PETSC_ERR_ARG_OUTOFRANGE
    b.MatOpCheckOperatorBlocks(OpSz, 0)
    b.MatOpCheckOperatorBlocks(OpSp, 0)


def test_single_site_operators():
    """src/DMRGBlock.cpp:1131-1136 (Sz = diag(+1/2,-1/2)), :1193-1195 (Sp = |0><1|), sectors {+1/2,-1/2}."""
    s = Block.single_site()
    assert np.array_equal(s.Sz(0).toarray(), np.diag([0.5, -0.5]))
    assert np.array_equal(s.Sp(0).toarray(), np.array([[0.0, 1.0], [0.0, 0.0]]))
    assert s.Magnetization.qn_list == [0.5, -0.5] and s.Magnetization.qn_size == [1, 1]
    s.CheckOperatorBlocks()


def test_quantum_numbers_ranges():
    q = QuantumNumbers([1.5, 0.5, -0.5, -1.5], [2, 3, 2, 1])
    assert q.Offsets() == [0, 2, 5, 7, 8]
    assert q.OpBlockToGlobalRange(1, OpSp) == (5, 7, True)
    assert q.OpBlockToGlobalRange(3, OpSp)[2] is False and q.OpBlockToGlobalRange(0, OpSm)[2] is False
    assert q.GlobalIdxToBlockIdx(6) == (2, 1)
    with pytest.raises(OracleError):
        QuantumNumbers([0.5, 0.5], [1, 1])


def test_hamiltonian_terms_and_quirks():
    """Bond counts of SURVEY section 6 and the reference's quirks (src/Hamiltonians.cpp:32-37,101)."""
    def nbonds(**kw):
        H = J1J2XXZModel_SquareLattice(**kw)
        return len({(t.Isite, t.Jsite, t.a) for t in H.H() if t.Iop == OpSp})
    assert nbonds(Lx=16, Ly=1, heisenberg=1.0) == 15
    assert len([t for t in J1J2XXZModel_SquareLattice(Lx=4, Ly=2, heisenberg=1.0).H() if t.Iop == OpSp]) == 14   # doubled vertical bonds
    assert len([t for t in J1J2XXZModel_SquareLattice(Lx=4, Ly=4, heisenberg=1.0).H() if t.Iop == OpSp]) == 28
    assert len([t for t in J1J2XXZModel_SquareLattice(Lx=4, Ly=4, J1=1, Jz1=1, J2=0.5, Jz2=0.5).H() if t.Iop == OpSp]) == 52
    assert len([t for t in J1J2XXZModel_SquareLattice(Lx=4, Ly=4).H() if t.Iop == OpSp]) == 28      # J2=1,Jz2=0: NNN dropped
    H = J1J2XXZModel_SquareLattice(Lx=5, Ly=4)
    for idx in range(20):
        assert H.To1D(*H.To2D(idx)) == idx
    assert [H.To1D(1, j) for j in range(4)] == [7, 6, 5, 4]        # S-snake: odd columns run downwards


def test_warmup_schedule_matches_survey_examples():
    """include/DMRGBlockContainer.hpp:809-840: Ly=4: 4->2, 5->5, 6->4, 7->7, 8->6 ; Ly=1: 2->2, 3->3, 4->4."""
    d = DMRGOracle(J1J2XXZModel_SquareLattice(Lx=6, Ly=4, heisenberg=1.0), 8)
    assert d.warmup_schedule()[1][:5] == [(4, 2), (5, 5), (6, 4), (7, 7), (8, 6)]
    d = DMRGOracle(J1J2XXZModel_SquareLattice(Lx=16, Ly=1, heisenberg=1.0), 8)
    assert d.warmup_schedule() == (2, [(2, 2), (3, 3), (4, 4), (5, 5), (6, 6), (7, 7)])


def _two_block_problem(seed=0):
    H = J1J2XXZModel_SquareLattice(Lx=4, Ly=2, J1=1.0, Jz1=0.7, J2=0.4, Jz2=0.3)
    site = Block.single_site()
    blk = site
    for n in range(2, 5):
        blk = KronEye_Explicit(blk, site, H.H(n))
    return H, blk


def test_shell_matvec_three_statements_agree():
    """Literal row loop (python and C, src/DMRGKron.cpp:1842-1864) == explicit assembly (KronSumFillMatrix, :1340-1477)."""
    H, blk = _two_block_problem()
    kb = KronBlocks(blk, blk, (0.0,))
    terms = H.H(8)
    Hx = KronSumConstruct_explicit(kb, terms)
    assert abs(Hx - Hx.T).max() < 1e-14
    sh = ShellCtx(kb, terms)
    x = np.random.default_rng(1).standard_normal(kb.NumStates())
    y_exp = Hx @ x
    assert np.abs(sh.apply_literal(x) - y_exp).max() < 1e-13
    c = ShellApplyC(sh)
    assert np.abs(c.apply(x) - y_exp).max() < 1e-13
    assert np.abs(c.apply(x, nthreads=3) - y_exp).max() < 1e-13            # cost-balanced row ranges
    part = c.apply(x, 5, 40)
    assert np.abs(part[5:40] - y_exp[5:40]).max() < 1e-13 and not part[:5].any() and not part[40:].any()


def test_exact_energy_small_lattice_and_truncation_invariants():
    """4x2 Heisenberg (doubled vertical bonds): E0 = -6.6682766346354 (ED); RDM invariants of SURVEY 8c."""
    H = J1J2XXZModel_SquareLattice(Lx=4, Ly=2, heisenberg=1.0)
    d = DMRGOracle(H, 32)
    d.Warmup()
    d.Sweeps(nsweeps=1)
    e = min(s["GSEnergy"] for s in d.steps if s["NSites_SysEnl"] + s["NSites_EnvEnl"] == 8)
    assert abs(e - (-6.6682766346354)) < 1e-10 * abs(e)
    kb, psi = d.last["kb"], d.last["psi"]
    assert abs(np.dot(psi, psi) - 1.0) < 1e-12
    L, R = GetTruncation(kb, psi, 1000)
    sl = np.sort([v for _, v in L["spectra"]])[::-1]
    sr = np.sort([v for _, v in R["spectra"]])[::-1]
    assert abs(sl.sum() - 1.0) < 1e-12 and abs(sr.sum() - 1.0) < 1e-12
    k = min(len(sl), len(sr))
    assert np.abs(sl[:k] - sr[:k]).max() < 1e-12          # same non-zero spectrum on both sides
    assert abs(L["TruncErr"]) < 1e-12 and L["TruncErr"] > -1e-12
    L2, _ = GetTruncation(kb, psi, 3)
    assert L2["RotMatT"].shape[0] == 3 and 0.0 <= L2["TruncErr"] < 1.0
    U = L2["RotMatT"].toarray()
    assert np.abs(U @ U.T - np.eye(3)).max() < 1e-12


def test_exact_energy_chain_cfg1():
    """BASELINE config 1 lattice: 16x1 Heisenberg chain, m=64 -> E0 = -6.9117371455751 (ED), 2 sweeps."""
    H = J1J2XXZModel_SquareLattice(Lx=16, Ly=1, heisenberg=1.0)
    d = DMRGOracle(H, 64)
    d.Warmup()
    d.Sweeps(nsweeps=1)
    assert len([s for s in d.steps if s["loop"] == "Sweep"]) == 16 - 4          # N-4 steps per sweep
    e = min(s["GSEnergy"] for s in d.steps if s["NSites_SysEnl"] + s["NSites_EnvEnl"] == 16)
    assert abs(e - (-6.9117371455751)) < 1e-10 * abs(e)


def test_oracle_correlators_match_exact_diagonalisation():
    """Pins the oracle's restatement of the correlator path (SetUpCorrelation's reflection rule, operator products,
    <psi| P_sys (x) P_env |psi>; include/DMRGBlockContainer.hpp:627-682,2255-2410) against dense ED of the 4x2 lattice:
    with m = 64 nothing is truncated, so every correlator must equal the exact ground-state expectation value."""
    from helpers import lattice_ground_state
    from oracle.qn import OpSm, OpSz, OpSp
    ham = J1J2XXZModel_SquareLattice(Lx=4, Ly=2, heisenberg=1.0)
    e0, psi, site_op = lattice_ground_state(ham)
    orc = DMRGOracle(ham, 64)
    lists = [[(OpSz, 1)], [(OpSz, 6)]]
    for a, b in ham.NeighborPairs():
        lists += [[(OpSz, a), (OpSz, b)], [(OpSp, a), (OpSm, b)], [(OpSm, a), (OpSp, b)]]
    lists += [[(OpSz, ham.To1D(ix, 1)) for ix in range(4)], [(OpSz, 0), (OpSz, 1), (OpSz, 6), (OpSz, 7)],
              [(OpSp, 2), (OpSm, 3), (OpSz, 4)], [(OpSp, 5), (OpSm, 6)]]
    for l in lists:
        orc.SetUpCorrelation(l)
    orc.Warmup()
    orc.Sweeps(nsweeps=1)
    assert abs(orc.gse - e0) <= 1e-12 * abs(e0)
    assert len(orc.corr_values) == 2                      # end of warm-up, end of the sweep
    bond = 0.0
    for l, v in zip(lists, orc.corr_values[-1]):
        P = None
        for (op, i) in l:
            P = site_op(op, i) if P is None else P @ site_op(op, i)
        exact = float(psi @ (P @ psi))
        assert abs(v - exact) <= 1e-12, (l, v, exact)
    # bond energies add up to E0:  sum over terms a <O_i O_j>
    vals = dict((tuple(l), v) for l, v in zip(lists, orc.corr_values[-1]))
    for t in ham.H(8):
        bond += t.a * vals[((t.Iop, t.Isite), (t.Jop, t.Jsite))]
    assert abs(bond - e0) <= 1e-12 * abs(e0)


def test_matrix_free_superblock_operator_equals_the_explicit_matrix():
    """oracle/kron.py: KronSumOperator (what the oracle solves superblocks above `matrix_free_above` states with: the golden table of
    tests/golden/make_engine_golden_large_m.py) against KronSumConstruct_explicit (src/DMRGKron.cpp:1340-1477) on a truncated Ly = 4 J1-J2 run
    with NNN terms: same matrix-vector product to round-off, same ground state, and the step records of a sweep solved matrix-free equal those
    of the same sweep solved on the explicit matrix."""
    import copy
    import numpy as np
    from oracle.kron import KronSumOperator, KronSumConstruct_explicit
    from oracle.dmrg import lowest_eigenpair
    ham = J1J2XXZModel_SquareLattice(Lx=6, Ly=4, J1=1.0, Jz1=0.8, J2=0.5, Jz2=0.3)
    orc = DMRGOracle(ham, 6, qn_sector=1.0)          # (the recipe of tests/golden/engine_medium_m.json's j1j2_6x4_sz1: every cut well-defined)
    orc.Warmup()
    for m in (8, 12):
        orc.SingleSweep(m, min_block=4)
    kb, terms = orc.last["kb"], orc.last["Terms"]
    He, Hm = KronSumConstruct_explicit(kb, terms), KronSumOperator(kb, terms)
    rng = np.random.default_rng(3)
    for _ in range(3):
        x = rng.standard_normal(kb.NumStates())
        y = He @ x
        assert np.abs(Hm.matvec(x) - y).max() <= 1e-14 * np.abs(y).max()
    H = np.stack([Hm.matvec(e) for e in np.eye(kb.NumStates())], axis=1)
    assert np.abs(H - H.T).max() <= 1e-14 * np.abs(H).max() and np.abs(H - He.toarray()).max() <= 1e-14 * np.abs(H).max()
    if kb.NumStates() > 1500:
        e1, v1 = lowest_eigenpair(He, seed=5)
        e2, v2 = lowest_eigenpair(Hm, seed=5)
        assert abs(e1 - e2) <= 1e-12 * abs(e1) and abs(abs(v1 @ v2) - 1.0) <= 1e-10
    a, b = copy.deepcopy(orc), copy.deepcopy(orc)
    b.matrix_free_above = 0
    a.SingleSweep(18, min_block=4); b.SingleSweep(18, min_block=4)
    n0 = len(orc.steps)
    for sa, sb_ in zip(a.steps[n0:], b.steps[n0:]):
        assert sa["NumStates_H"] == sb_["NumStates_H"] and sa["NStates_SysRot"] == sb_["NStates_SysRot"]
        assert abs(sa["GSEnergy"] - sb_["GSEnergy"]) <= 1e-11 * abs(sa["GSEnergy"])
        assert abs(sa["TruncErr_Sys"] - sb_["TruncErr_Sys"]) <= 1e-9 * abs(sa["TruncErr_Sys"]) + 1e-14
