#!/usr/bin/env python3
"""Golden step tables for the headline geometries (BASELINE configs[2..4]: Heisenberg 16x6, J1-J2 20x8, XY 32x8) at reduced m.

The CPU oracle (oracle/dmrg.py, this repository's restatement of include/DMRGBlockContainer.hpp:687-2057 and
src/Hamiltonians.cpp:70-122) needs 1-5 minutes of single-threaded Python per lattice, too long for the GPU test tier, so its step
records are generated here once and committed as tests/golden/engine_big_lattices.json; tests/test_gpu_engine.py compares the
engine with them step by step.  For every lattice a few (m, Sz sector) candidates are scanned and the one with the fewest
ill-defined cuts is kept (a cut is well-defined when the last kept eigenvalue is above round-off and separated from the first
dropped one: only then is the kept subspace -- and everything after it -- determined to round-off in ANY implementation).

    python tests/golden/make_engine_golden.py          # ~10 minutes on 8 cores
"""
import json, os, sys, time
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASES = {
    "cfg4_j1j2_20x8": dict(Lx=20, Ly=8, J1=1.0, Jz1=1.0, J2=0.5, Jz2=0.5, cand=[(8, 1), (6, 1), (10, 1), (8, 2)]),
    "cfg3_heisenberg_16x6": dict(Lx=16, Ly=6, J1=0.5, Jz1=1.0, J2=0.0, Jz2=0.0, cand=[(4, 2), (6, 1), (10, 1), (8, 2), (12, 1)]),   # (4, 2): every one of its 134 cuts is well-defined
    "cfg5_xy_32x8": dict(Lx=32, Ly=8, J1=1.0, Jz1=0.0, J2=1.0, Jz2=0.0, cand=[(8, 1), (6, 1)]),      # Jz2 = 0: the reference drops the NNN bonds
}
KEYS = ("NSites_Sys", "NSites_Env", "NStates_SysEnl", "NStates_EnvEnl", "NumStates_H", "NStates_SysRot", "NStates_EnvRot", "GSEnergy", "TruncErr_Sys", "TruncErr_Env")


def run(args):
    name, m, sz = args
    from oracle.hamiltonian import J1J2XXZModel_SquareLattice
    from oracle.dmrg import DMRGOracle
    c = CASES[name]
    H = J1J2XXZModel_SquareLattice(Lx=c["Lx"], Ly=c["Ly"], J1=c["J1"], Jz1=c["Jz1"], J2=c["J2"], Jz2=c["Jz2"])
    t0 = time.time()
    o = DMRGOracle(H, m, qn_sector=float(sz))
    o.Warmup()
    o.Sweeps(nsweeps=1)
    steps = []
    for s in o.steps:
        ok = all(lk > 1e-9 and (ld == 0.0 or (lk - ld) / lk > 1e-2) for lk, ld in (s["cut_Sys"], s["cut_Env"]))
        steps.append(dict({k: (int(s[k]) if k.startswith("N") else float(s[k])) for k in KEYS}, well_defined=bool(ok), nterms=int(s["nterms"])))
    return dict(name=name, m=m, qn_sector=sz, nsweeps=1, steps=steps, seconds=time.time() - t0, n_ill=sum(not s["well_defined"] for s in steps),
                options=dict(Lx=c["Lx"], Ly=c["Ly"], J1=c["J1"], Jz1=c["Jz1"], J2=c["J2"], Jz2=c["Jz2"]), nterms_full=len(H.H(c["Lx"] * c["Ly"])))


if __name__ == "__main__":
    jobs = [(n, m, sz) for n, c in CASES.items() for (m, sz) in c["cand"]]
    best = {}
    with ProcessPoolExecutor(max_workers=min(len(jobs), 7)) as ex:
        for r in ex.map(run, jobs):
            first_ill = next((i for i, s in enumerate(r["steps"]) if not s["well_defined"]), len(r["steps"]))
            print(f"{r['name']} m={r['m']} sz={r['qn_sector']}: {len(r['steps'])} steps, {r['n_ill']} ill-defined cuts (first at step {first_ill}), {r['seconds']:.0f} s", flush=True)
            key = (r["n_ill"], -first_ill)
            if r["name"] not in best or key < best[r["name"]][0]:
                best[r["name"]] = (key, r)
    out = {n: b[1] for n, b in best.items()}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "engine_big_lattices.json"), "w"), indent=0)
    for n, r in out.items():
        print("kept:", n, "m", r["m"], "sz", r["qn_sector"], "ill", r["n_ill"], "E", r["steps"][-1]["GSEnergy"])
