#!/usr/bin/env python3
"""Transcribe the known-answer tables held by the reference's own tests into JSON fixtures.

Reads the reference test sources AS TEXT (they cannot be built here: PETSc/SLEPc absent) and extracts only the
numeric tables -- sector lists, row patterns and expected rows:
  * tests/UnitTests_DMRGKron.cpp:49-95 (inputs) and :117-245 (expected rows)  -> testkron01.json
  * tests/UnitTests_Misc.cpp:82-136 (SetSz0/SetSp0/SetSz1/SetSp1 patterns)     -> block_fixture.json
  * tests/UnitTests_DMRGBlock.cpp:84-114 (valid rows + the planted bad row)    -> block_fixture.json
SetRow stores value == column index (tests/UnitTests_Misc.cpp:17), which is what "inputs" encodes.
Run once in the build container:  python tests/golden/make_golden_from_reference_tests.py
"""
import json, os, re, sys

REF = "/root/reference/tests"
HERE = os.path.dirname(os.path.abspath(__file__))


def ints(s):
    return [int(x) for x in re.findall(r"-?\d+", s)]


def floats(s):
    return [float(x) for x in re.findall(r"[-+]?\d*\.?\d+", s)]


def kron01():
    src = open(os.path.join(REF, "UnitTests_DMRGKron.cpp")).read()
    body = src[src.index("PetscErrorCode TestKron01()\n{"):src.index("PetscErrorCode TestKron02()\n{")]
    blocks = {}
    for m in re.finditer(r"(\w+)\.Initialize\(PETSC_COMM_WORLD,\s*(\d+),\s*\{([^}]*)\},\s*\{([^}]*)\}\)", body):
        blocks[m.group(1)] = dict(nsites=int(m.group(2)), qn_list=floats(m.group(3)), qn_size=ints(m.group(4)), Sz={}, Sp={})
    for m in re.finditer(r"SetRow\(\s*(\w+)\.(Sz|Sp)\((\d+)\),\s*(\d+),\s*\{([^}]*)\}\)", body):
        blk, op, site, row, cols = m.group(1), m.group(2), m.group(3), m.group(4), ints(m.group(5))
        blocks[blk][op].setdefault(site, {})[row] = cols
    expected = {"Sz": {}, "Sp": {}}
    for m in re.finditer(r'CheckRow\(BlockOut\.(Sz|Sp)\((\d+)\),\s*"[^"]*",\s*(\d+),\s*\{([^}]*)\},\s*\{([^}]*)\}\)', body):
        op, site, row = m.group(1), m.group(2), m.group(3)
        expected[op].setdefault(site, {})[row] = dict(cols=ints(m.group(4)), vals=floats(m.group(5)))
    n = sum(len(r) for op in expected.values() for r in op.values())
    assert n == 120, n  # 10 operators x 12 rows
    return dict(source="tests/UnitTests_DMRGKron.cpp:39-252 (TestKron01)", left=blocks["LeftBlock"],
                right=blocks["RightBlock"], expected=expected)


def block_fixture():
    src = open(os.path.join(REF, "UnitTests_Misc.cpp")).read()
    pats = {}
    for name in ("SetSz0", "SetSp0", "SetSz1", "SetSp1"):
        body = src[src.index(f"PetscErrorCode {name}("):]
        body = body[:body.index("return ierr;")]
        pats[name] = {m.group(1): ints(m.group(2)) for m in re.finditer(r"SetRow\(\w+,\s*(\d+),\s*\{([^}]*)\}\)", body)}
    src2 = open(os.path.join(REF, "UnitTests_DMRGBlock.cpp")).read()
    body = src2[src2.index("PetscErrorCode Test_MatOpCheckOperatorBlocks()"):src2.index("PetscErrorCode Test_SavingBlocks()")]
    m = re.search(r"Initialize\(PETSC_COMM_WORLD,\s*(\d+),\s*\{([^}]*)\},\s*\{([^}]*)\}\)", body)
    planted = {}
    for mm in re.finditer(r"SetRow\(blk\.(Sz|Sp)\((\d+)\),\s*(\d+),\s*\{([^}]*)\}\)", body):
        planted.setdefault(f"{mm.group(1)}{mm.group(2)}", {})[mm.group(3)] = ints(mm.group(4))
    return dict(source="tests/UnitTests_Misc.cpp:82-136, tests/UnitTests_DMRGBlock.cpp:76-131",
                nsites=int(m.group(1)), qn_list=floats(m.group(2)), qn_size=ints(m.group(3)),
                valid=pats, planted=planted, planted_bad=dict(op="Sz", site=1, row=7, expect_code=63,
                                                              expect_code_name="PETSC_ERR_ARG_OUTOFRANGE"))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures are committed, nothing to do")
    json.dump(kron01(), open(os.path.join(HERE, "testkron01.json"), "w"), indent=1, sort_keys=True)
    json.dump(block_fixture(), open(os.path.join(HERE, "block_fixture.json"), "w"), indent=1, sort_keys=True)
    print("wrote testkron01.json, block_fixture.json")
