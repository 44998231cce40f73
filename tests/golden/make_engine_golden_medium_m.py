#!/usr/bin/env python3
"""Golden step tables at m = 24 ... 48: the sizes at which the engine's RDM solver merges divide-and-conquer sub-problems with
deflation, back-transforms through WY blocks and the GEMMs use more than one 16 x 16 MFMA block per sector.

Why these runs look the way they do.  A step can only be compared at 1e-10 if its m-cut is well-defined: the last kept eigenvalue
of the density matrix above round-off and separated from the first dropped one -- else the kept subspace, and every later energy, is
decided by rounding noise in ANY implementation (the reference's included).  At m >= 24 that rules out
  * the reference's plain schedule: the first warm-up steps and the last steps of every sweep truncate a large block against an
    environment of two or three sites, whose density matrix has rank <= 8 -- so the runs warm up at m = 6, grow m by ~1.45 x per
    sweep (the rank available to a sweep is set by the previous sweep's m) and turn round `min_block` sites before the edge (the
    MinBlock argument of the reference's SingleSweep, include/DMRGBlockContainer.hpp:996-1013; `-min_block` on the engine's command
    line), where the environment is still an exactly-kept block;
  * SU(2)-symmetric couplings in the Sz = 0 sector (multiplets at the cut): anisotropic couplings / Sz = 1.
The per-sweep m of every case below was found by the search mode of this script (--search: for each sweep up to eight candidates
around 1.45 x the previous m, the first one whose sweep has only well-defined cuts; ~1 hour on 6 cores for the 30 (lattice, Sz,
min_block) combinations tried).  Without --search the four kept cases are re-run as recorded (~15 minutes on 4 cores) and written to
tests/golden/engine_medium_m.json; tests/test_gpu_engine.py::test_medium_m_step_by_step_against_the_oracle compares the engine with it.
"""
import copy, json, os, sys, time
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

# name: lattice, couplings, Sz sector, min_block, warm-up m, m of every sweep
CASES = {
    "j1j2_10x4_sz1": dict(Lx=10, Ly=4, J1=1.0, Jz1=0.8, J2=0.5, Jz2=0.3, qn_sector=1, min_block=4, mwarmup=6, msweeps=[8, 12, 16, 24, 32, 46]),
    "j1j2_8x4_sz1": dict(Lx=8, Ly=4, J1=1.0, Jz1=1.0, J2=0.5, Jz2=0.5, qn_sector=1, min_block=4, mwarmup=6, msweeps=[10, 14, 20, 28, 41]),
    "xxz_8x6_sz1": dict(Lx=8, Ly=6, J1=0.5, Jz1=0.7, J2=0.0, Jz2=0.0, qn_sector=1, min_block=6, mwarmup=6, msweeps=[8, 12, 16, 25]),
    "j1j2_6x4_sz1": dict(Lx=6, Ly=4, J1=1.0, Jz1=0.8, J2=0.5, Jz2=0.3, qn_sector=1, min_block=4, mwarmup=6, msweeps=[8, 12, 18, 26, 38]),
}
KEYS = ("NSites_Sys", "NSites_Env", "NStates_SysEnl", "NStates_EnvEnl", "NumStates_H", "NStates_SysRot", "NStates_EnvRot", "GSEnergy", "TruncErr_Sys", "TruncErr_Env")


def well_defined(s):
    """every m-cut of the step is decided by the spectrum, not by round-off (nothing dropped counts as decided)"""
    ok = True
    for side, (lk, ld) in (("Sys", s["cut_Sys"]), ("Env", s["cut_Env"])):
        full = s["NStates_%sRot" % side] == s["NStates_%sEnl" % side]
        ok = ok and (full or (lk > 1e-9 and (lk - ld) / lk > 1e-2))
    return bool(ok)


def oracle_for(c, m0):
    from oracle.hamiltonian import J1J2XXZModel_SquareLattice
    from oracle.dmrg import DMRGOracle
    H = J1J2XXZModel_SquareLattice(Lx=c["Lx"], Ly=c["Ly"], J1=c["J1"], Jz1=c["Jz1"], J2=c["J2"], Jz2=c["Jz2"])
    return DMRGOracle(H, m0, qn_sector=float(c["qn_sector"]))


def run(name):
    c = CASES[name]
    t0 = time.time()
    o = oracle_for(c, c["mwarmup"])
    o.Warmup()
    m_of_step = [c["mwarmup"]] * len(o.steps)
    for m in c["msweeps"]:
        n0 = len(o.steps)
        o.SingleSweep(m, min_block=c["min_block"])
        m_of_step += [m] * (len(o.steps) - n0)
    steps = [dict({k: (int(s[k]) if k.startswith("N") else float(s[k])) for k in KEYS}, well_defined=well_defined(s), m=int(m),
                  max_sector=int(max(s["sectors_SysEnl"][1]))) for s, m in zip(o.steps, m_of_step)]
    first_ill = next((i for i, s in enumerate(steps) if not s["well_defined"]), len(steps))
    return name, dict(options={k: c[k] for k in ("Lx", "Ly", "J1", "Jz1", "J2", "Jz2")}, qn_sector=c["qn_sector"], min_block=c["min_block"], mwarmup=c["mwarmup"],
                      msweeps=c["msweeps"], steps=steps, first_ill=first_ill, strict_steps_m24=sum(1 for s in steps[:first_ill] if s["m"] >= 24),
                      seconds=time.time() - t0)


def search(args):
    """(Lx, Ly, J1, Jz1, J2, Jz2, Sz, min_block, top m): warm-up m and per-sweep m with only well-defined cuts, as far as they exist"""
    Lx, Ly, J1, Jz1, J2, Jz2, sz, mb, top = args
    c = dict(Lx=Lx, Ly=Ly, J1=J1, Jz1=Jz1, J2=J2, Jz2=Jz2, qn_sector=sz)
    o = None
    for m0 in (6, 5, 7, 8, 4, 10):
        o = oracle_for(c, m0)
        o.Warmup()
        if all(well_defined(s) for s in o.steps):
            break
    else:
        return f"{args}: no well-defined warm-up"
    ms, m, strict = [], m0, True
    while m < top and strict:
        cands = sorted(range(max(m + 1, int(m * 1.25)), int(m * 1.7) + 1), key=lambda x: abs(x - m * 1.45))[:8]
        found, best = None, None
        for cm in cands:
            o2 = copy.deepcopy(o)
            n0 = len(o2.steps)
            o2.SingleSweep(cm, min_block=mb)
            oks = [well_defined(s) for s in o2.steps[n0:]]
            if all(oks):
                found = (cm, o2)
                break
            if best is None or oks.index(False) > best[0]:
                best = (oks.index(False), cm, o2)
        if found:
            m, o = found
        else:
            _, m, o = best
            strict = False
        ms.append(m)
    first_ill = next((i for i, s in enumerate(o.steps) if not well_defined(s)), len(o.steps))
    return f"{args}: mwarmup {m0} msweeps {ms} steps {len(o.steps)} first ill-defined cut at step {first_ill}"


if __name__ == "__main__":
    if "--search" in sys.argv:
        jobs = [(Lx, Ly, J1, Jz1, J2, Jz2, sz, mb, 64) for (Lx, Ly, J1, Jz1, J2, Jz2) in ((6, 4, 1.0, 0.8, 0.5, 0.3), (8, 4, 1.0, 0.8, 0.5, 0.3), (8, 6, 0.5, 0.7, 0.0, 0.0), (8, 4, 1.0, 1.0, 0.5, 0.5), (10, 4, 1.0, 0.8, 0.5, 0.3))
                for sz in (1, 2) for mb in (4, 5, 6)]
        with ProcessPoolExecutor(max_workers=6) as ex:
            for r in ex.map(search, jobs):
                print(r, flush=True)
        sys.exit(0)
    out = {}
    with ProcessPoolExecutor(max_workers=4) as ex:
        for name, r in ex.map(run, list(CASES)):
            out[name] = r
            print(f"{name}: {len(r['steps'])} steps, first ill-defined cut at step {r['first_ill']}, strict steps at m >= 24: {r['strict_steps_m24']}, "
                  f"largest sector {max(s['max_sector'] for s in r['steps'])}, E {r['steps'][-1]['GSEnergy']:.12f}, {r['seconds']:.0f} s", flush=True)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "engine_medium_m.json"), "w"), indent=0)
