"""CPU suite: host logic of the C++ engine (dmrg.x_amd/host/*.hpp) replayed through dmrgx-host-tool -- no GPU involved.
The reference's known-answer tables are applied to the ENGINE here (tests/test_oracle_golden.py applies them to the oracle)."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle.hamiltonian import J1J2XXZModel_SquareLattice

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "dmrg.x_amd", "dmrgx-host-tool")
GOLD = os.path.join(ROOT, "tests", "golden")


def tool(lines):
    out = subprocess.run([TOOL], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout.splitlines()


def block_lines(name, d):
    L = [f"block {name} {d['nsites']} {len(d['qn_list'])} " + " ".join(map(str, d["qn_list"])) + " " + " ".join(map(str, d["qn_size"]))]
    for op in ("Sz", "Sp"):
        for site, rows in d[op].items():
            for row, cols in rows.items():
                for c in cols:
                    L.append(f"set {name} {op} {site} {row} {c} {float(c)}")      # SetRow: value == column index
    return L


def parse_dump(lines):
    ops, cur = {}, None
    for ln in lines:
        t = ln.split()
        if t[0] == "op":
            cur = (t[1], int(t[2]))
            ops[cur] = {}
        elif t[0] == "row" and cur:
            ops[cur][int(t[1])] = {int(a.split(":")[0]): float(a.split(":")[1]) for a in t[2:]}
    return ops


def test_engine_kroneye_matches_reference_table():
    """tests/UnitTests_DMRGKron.cpp:39-252 on the engine's KronEye_Explicit (sector merging + index formula)."""
    g = json.load(open(os.path.join(GOLD, "testkron01.json")))
    out = tool(block_lines("L", g["left"]) + block_lines("R", g["right"]) + ["kroneye L R O", "check O", "dump O"])
    rcs = [ln for ln in out if ln.startswith("rc ")]
    assert all(ln == "rc 0" for ln in rcs), rcs
    sect = next(ln for ln in out if ln.startswith("sectors")).split()
    assert sect[1:] == ["4", "1.5", "0.5", "-0.5", "-1.5", "2", "5", "4", "1"]
    ops = parse_dump(out)
    n = 0
    for opname in ("Sz", "Sp"):
        for site, rows in g["expected"][opname].items():
            for row, exp in rows.items():
                want = {c: v for c, v in zip(exp["cols"], exp["vals"]) if v != 0.0}     # dense cells carry no structural zeros
                assert ops[(opname, int(site))].get(int(row), {}) == want, (opname, site, row)
                n += 1
    assert n == 120


def test_kronblocks_iterator_decodes_rows_like_the_reference():
    """SURVEY 8a row A2 (reference include/DMRGKron.hpp:501-656): KronBlocksIterator walks superblock rows and decodes
    (KronBlock, local index, left/right sector, left/right local and global index).  Checked three ways on the blocks of
    the reference's TestKron01: against the oracle's restatement of the decode (:587-620), against the reference-held
    expected table (a left-site operator O (x) 1 has out[r, r'] = O[gL(r), gL(r')] delta(gR(r), gR(r')), so the table's
    column lists follow from the decoded (gL, gR) alone), and for a sub-range [istart, iend) that starts inside a block."""
    from oracle.block import Block as OBlock
    from oracle.kron import KronBlocks as OKronBlocks
    g = json.load(open(os.path.join(GOLD, "testkron01.json")))
    out = tool(block_lines("L", g["left"]) + block_lines("R", g["right"]) + ["iterate L R 0 -1", "iterate L R 5 11", "iterate L R 0 -1 0.5"])
    runs, cur = [], None
    for ln in out:
        t = ln.split()
        if t[0] == "iterate":
            cur = {"range": (int(t[1]), int(t[2])), "rows": []}
            runs.append(cur)
        elif t[0] == "it":
            cur["rows"].append([int(v) for v in t[1:]])
    full, part, sector = runs
    def oblock(d):
        return OBlock.with_sectors(d["nsites"], d["qn_list"], d["qn_size"])
    L, R = oblock(g["left"]), oblock(g["right"])
    for run, qn in ((full, ()), (sector, (0.5,))):
        kb = OKronBlocks(L, R, qn)
        k, IL, IR, lL, lR, gL, gR = kb.rows()
        assert run["range"] == (0, kb.NumStates()) and len(run["rows"]) == kb.NumStates() > 0
        offs = [kb.Offsets(i) for i in range(kb.size() + 1)]
        for idx, row in enumerate(run["rows"]):
            assert row[:9] == [idx, k[idx], idx - offs[k[idx]], IL[idx], IR[idx], lL[idx], lR[idx], gL[idx], gR[idx]], idx
            assert row[9] == int(idx == 0 or k[idx] != k[idx - 1])                              # UpdatedBlock
            assert row[10] == idx and row[11] == offs[k[idx]]                                   # Steps, BlockStartIdx(0)
            assert row[12] == (kb.kb[k[idx] + 1][3] if k[idx] + 1 < kb.size() else -1)          # BlockSize(+1), -1 past the end
    assert part["range"] == (5, 11) and [r[0] for r in part["rows"]] == list(range(5, 11))
    assert [r[1:9] for r in part["rows"]] == [r[1:9] for r in full["rows"][5:11]] and [r[10] for r in part["rows"]] == list(range(6))
    # the reference's expected table, re-derived from the decoded indices
    gLs, gRs = [r[7] for r in full["rows"]], [r[8] for r in full["rows"]]
    nl, checked = g["left"]["nsites"], 0
    for opname in ("Sz", "Sp"):
        for site, rows in g["expected"][opname].items():
            if int(site) >= nl:
                continue
            src = g["left"][opname][site]                                                       # row -> columns (value == column index)
            for row, exp in rows.items():
                r = int(row)
                want = {c: v for c, v in zip(exp["cols"], exp["vals"]) if v != 0.0}
                got = {rp: float(gLs[rp]) for rp in range(len(gLs)) if gRs[rp] == gRs[r] and gLs[rp] in src.get(str(gLs[r]), []) and gLs[rp] != 0}
                assert got == want, (opname, site, row)
                checked += 1
    assert checked >= 40


def test_engine_block_checks_and_planted_error():
    """tests/UnitTests_DMRGBlock.cpp:76-131: valid patterns pass; an entry outside its sector block is refused with
    PETSC_ERR_ARG_OUTOFRANGE (63) -- in the engine at insertion time, since cells cannot hold such an entry."""
    f = json.load(open(os.path.join(GOLD, "block_fixture.json")))
    d = dict(nsites=f["nsites"], qn_list=f["qn_list"], qn_size=f["qn_size"], Sz={"0": f["valid"]["SetSz0"], "1": f["valid"]["SetSz1"]},
             Sp={"0": f["valid"]["SetSp0"], "1": f["valid"]["SetSp1"]})
    out = tool(block_lines("B", d) + ["check B"])
    assert all(ln == "rc 0" for ln in out if ln.startswith("rc "))
    bad = f["planted_bad"]
    out = tool([f"block B {f['nsites']} 4 " + " ".join(map(str, f["qn_list"])) + " " + " ".join(map(str, f["qn_size"])),
                f"set B Sz {bad['site']} {bad['row']} 1 1.0", "set B Sz 1 7 7 7.0", "qnrange B 3 1", "qnrange B 1 1"])
    assert out[1] == f"rc {bad['expect_code']}" and out[2] == "rc 0"
    assert out[3].split()[-1] == "0" and out[4] == "rc 0 5 7 1"


def test_engine_single_site_and_kronblocks_order():
    # (enlarging blocks that carry a Hamiltonian needs the device: covered by tests/test_gpu_engine.py)
    out = tool(["single A", "single B", "dump A", "kronblocks A B", "kronblocks A B 0",
                "block P 2 2 0.5 -0.5 1 1", "block Q 2 2 0.5 -0.5 1 1", "kroneye P Q C", "dump C"])
    ops = parse_dump(out[:out.index("end") + 1])                 # the dump of A (the dump of C follows later)
    assert ops[("Sz", 0)] == {0: {0: 0.5}, 1: {1: -0.5}} and ops[("Sp", 0)] == {0: {1: 1.0}, 1: {}}
    kb_all = next(ln for ln in out if ln.startswith("kronblocks 4"))
    assert kb_all.split()[3:] == ["0,0,1,0", "0,1,1,1", "1,0,1,2", "1,1,1,3"]          # stable sort by descending Sz
    kb0 = [ln for ln in out if ln.startswith("kronblocks 2")][0]
    assert kb0.split()[3:] == ["0,1,1,0", "1,0,1,1"]                                    # target Sz=0: nested IL-then-IR order
    sect = [ln for ln in out if ln.startswith("sectors")][-1].split()
    assert sect[1:] == ["3", "1", "0", "-1", "1", "2", "1"]


@pytest.mark.parametrize("opts", [dict(Lx=16, Ly=1, heisenberg=1.0), dict(Lx=4, Ly=2, heisenberg=1.0), dict(Lx=4, Ly=4, J1=1, Jz1=1, J2=0.5, Jz2=0.5),
                                  dict(Lx=4, Ly=4), dict(Lx=6, Ly=3, J1=1, Jz1=0.3, J2=0.5, Jz2=0.2, BCperiodic=True), dict(Lx=5, Ly=4, J1=1, Jz1=1, J2=1, Jz2=1, BCopen=True),
                                  # the headline geometries of BASELINE configs[3], [2], [4]: width-8 and width-6 cylinders, and the XY point where
                                  # the reference drops the J2 bonds together with Jz2 = 0 (src/Hamiltonians.cpp:101)
                                  dict(Lx=20, Ly=8, J1=1, Jz1=1, J2=0.5, Jz2=0.5), dict(Lx=16, Ly=6, heisenberg=1.0), dict(Lx=32, Ly=8, J1=1, Jz1=0, J2=1, Jz2=0)])
def test_engine_hamiltonian_terms_match_oracle(opts):
    """Hamiltonians::J1J2XXZModel_SquareLattice::H(n) of the engine == the oracle's restatement of
    src/Hamiltonians.cpp:70-122, term by term and in order, for full and partial lattices."""
    H = J1J2XXZModel_SquareLattice(**opts)
    args = []
    for k, v in opts.items():
        args += [f"-{k}", "_" if v is True else str(v)]
    ns = H.NumSites()
    out = tool(["ham " + " ".join(args), "terms -1", f"terms {ns // 2}", f"terms {ns - 3}", "snake"])
    for ln, n in zip(out[1:4], (None, ns // 2, ns - 3)):
        got = [tuple(float(x) if i == 0 else int(x) for i, x in enumerate(t.split(","))) for t in ln.split()[2:]]
        want = [(t.a, t.Iop, t.Isite, t.Jop, t.Jsite) for t in H.H(n)]
        assert got == want
    snake = [tuple(int(x) for x in t.split(",")) for t in out[4].split()[1:]]
    assert all(H.To2D(i) == (ix, jy) and idx == i for i, (ix, jy, idx) in enumerate(snake))


def test_engine_block_checkpoint_round_trip(tmp_path):
    """tests/UnitTests_DMRGBlock_SaveInfo.cpp:17-88 of the reference: a block written to disk (BlockInfo.dat,
    QuantumNumbers.dat, one file per operator) and rebuilt with InitializeFromDisk carries the same sectors and the same
    operator entries; a directory that does not exist and a tampered BlockInfo.dat are refused."""
    f = json.load(open(os.path.join(GOLD, "block_fixture.json")))
    d = dict(nsites=f["nsites"], qn_list=f["qn_list"], qn_size=f["qn_size"], Sz={"0": f["valid"]["SetSz0"], "1": f["valid"]["SetSz1"]},
             Sp={"0": f["valid"]["SetSp0"], "1": f["valid"]["SetSp1"]})
    d1, d2 = str(tmp_path / "blk"), str(tmp_path / "missing")
    os.makedirs(d1)
    out = tool(block_lines("B", d) + [f"save B {d1}", f"save B {d2}", "dump B", f"load C {d1}", "check C", "dump C"])
    rcs = [ln for ln in out if ln.startswith("rc ")]
    n_set = len(block_lines("B", d))
    assert rcs[n_set] == "rc 0" and rcs[n_set + 1] != "rc 0" and rcs[n_set + 2] == "rc 0" and rcs[n_set + 3] == "rc 0"
    ends = [i for i, ln in enumerate(out) if ln == "end"]
    first, second = out[:ends[0] + 1], out[ends[0] + 1:ends[1] + 1]
    s1 = [ln for ln in first if ln.startswith("sectors")][0]
    s2 = [ln for ln in second if ln.startswith("sectors")][0]
    assert s1 == s2
    assert parse_dump(first) == parse_dump(second) and len(parse_dump(first)) == 2 * f["nsites"]
    for name in ("BlockInfo.dat", "QuantumNumbers.dat", "Sz_000000000.mat", "Sp_000000001.mat"):
        assert os.path.exists(os.path.join(d1, name)), name
    info = open(os.path.join(d1, "BlockInfo.dat")).read().split()
    assert info[info.index("NumSites") + 1] == str(f["nsites"]) and info[info.index("NumSectors") + 1] == str(len(f["qn_list"]))
    # tamper: a different sector count must be caught by the cross-check
    txt = open(os.path.join(d1, "BlockInfo.dat")).read().replace("NumSectors" + " " * 21 + str(len(f["qn_list"])), "NumSectors" + " " * 21 + "9")
    open(os.path.join(d1, "BlockInfo.dat"), "w").write(txt)
    out = tool([f"load D {d1}"])
    assert out[-1] != "rc 0"


def test_engine_block_shallow_copy_semantics():
    """tests/UnitTests_DMRGBlock.cpp:26-73 of the reference: a copied block shares the operator handles of the original and
    passes the same checks; Destroy() on the copy releases the matrices of the original too."""
    f = json.load(open(os.path.join(GOLD, "block_fixture.json")))
    d = dict(nsites=f["nsites"], qn_list=f["qn_list"], qn_size=f["qn_size"], Sz={"0": f["valid"]["SetSz0"], "1": f["valid"]["SetSz1"]},
             Sp={"0": f["valid"]["SetSp0"], "1": f["valid"]["SetSp1"]})
    setup = block_lines("B", d)
    out = tool(setup + ["copy B C", "check C", "dump C", "nnz B 0", "destroy C", "nnz B 0"])
    rcs = [ln for ln in out if ln.startswith("rc ")]
    assert rcs[len(setup)] == "rc 0" and rcs[len(setup) + 1] == "rc 0"          # copy, check C
    first = tool(setup + ["dump B"])
    end = [i for i, ln in enumerate(out) if ln == "end"][0]
    assert parse_dump(out[:end + 1]) == parse_dump(first)                        # the copy shows the same operators
    nn = [ln for ln in out if ln.startswith("nnz ")]
    assert int(nn[0].split()[1]) > 0 and int(nn[1].split()[1]) <= 0             # ... and its Destroy() emptied the original's


def test_correlators_are_dealt_over_the_ranks_by_carried_weight():
    """Multi-rank runs deal the correlators over the ranks (host/CorrelatorDealing.hpp): every correlator has exactly one owner,
    a one-rank run keeps them all (-1), the one- and two-site correlators go to their lowest site's run, and the carried weight --
    sum over a rank's sites of the times each is rotated on the way back, N/2 - i for site i -- is balanced: on the driver's
    correlator table of the 20x8 cylinder (magnetisations, three bond correlators per nearest-neighbour pair mapped into the
    half-lattice block, one row string, two columns, one loop) no rank of 8 carries more than 34 % of what one rank carries alone
    (pairs reach Ly sites ahead and a string drags sites of the whole block along: 2.5 x redundancy in all), and the ranks are level."""
    Lx, Ly = 20, 8
    N = Lx * Ly
    H = N // 2
    local = lambda i: i if i < H else N - 1 - i          # (the right half is measured on the reflected block)
    corr = [[i] for i in range(H)]
    for x in range(Lx):
        for y in range(Ly):
            i = x * Ly + y
            for j in ([i + Ly] if x + 1 < Lx else []) + [x * Ly + (y + 1) % Ly]:
                corr += [[local(i), local(j)]] * 3
    corr.append([local(x * Ly + 1) for x in range(Lx)])                         # row string
    corr.append([local(1 * Ly + y) for y in range(Ly)])
    corr.append([local((Lx - 2) * Ly + y) for y in range(Ly)])
    loop = [1 * Ly + y for y in range(1, Ly - 2)] + [x * Ly + Ly - 2 for x in range(1, Lx - 2)] + [(Lx - 2) * Ly + y for y in range(Ly - 2, 1, -1)] + [x * Ly + 1 for x in range(Lx - 2, 1, -1)]
    corr.append([local(i) for i in loop])
    line = lambda W: "deal %d %d %d " % (N, W, len(corr)) + " ".join("%d %s" % (len(c), " ".join(map(str, c))) for c in corr)
    out = tool([line(1), line(2), line(8)])
    owners = [list(map(int, l.split()[1:])) for l in out if l.startswith("owners")]
    carried = [list(map(float, l.split()[1:])) for l in out if l.startswith("carried")]
    assert len(owners) == 3 and all(len(o) == len(corr) for o in owners)
    assert set(owners[0]) == {-1}
    weight = lambda sites: sum(max(H - i, 1) for i in set(sites))
    alone = weight([i for c in corr for i in c])
    for W, own, car in ((2, owners[1], carried[1]), (8, owners[2], carried[2])):
        assert set(own) == set(range(W))                                           # every rank measures something
        for r in range(W):                                                         # the tool's figure is the weight of the rank's site set
            assert abs(car[r] - weight([i for c, o in zip(corr, own) if o == r for i in c])) < 1e-9
        short = [(min(c), o) for c, o in zip(corr, own) if len(c) <= 2]
        assert all(o1 <= o2 for (l1, o1), (l2, o2) in zip(sorted(short), sorted(short)[1:]))      # runs of lowest sites, in order
        assert max(car) <= (0.72 if W == 2 else 0.34) * alone and max(car) <= 1.1 * min(car), (W, car, alone)
