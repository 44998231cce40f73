#!/usr/bin/env python3
"""Golden step table at m = 250 ... 480: the sizes at which the engine's density-matrix solver runs its
PRODUCTION paths inside a sweep -- matrices of order 140-260 dealt over several workgroups of the persistent tridiagonalisation
(`trid_coop_kernel`, csrc/symeig.hip), >= 3 divide-and-conquer merge levels, more than one compact-WY block in the back-transformation
-- and superblocks of 0.5-1.8 x 10^5 states (thousands of GEMM tiles per MatMult).

Same recipe as make_engine_golden_medium_m.py (warm-up at m = 6, m grown by ~1.45 x per sweep, `min_block` = 4: see its docstring),
one lattice: a J1-J2 4 x 8 cylinder (the Ly = 8 bond topology of the headline configuration), anisotropic couplings, Sz = 1, and
`min_block` = 8 = Ly: blocks of up to Ly sites are exact (include/DMRGBlockContainer.hpp:786-790), so the block the sweep turns round on
has 256 states and the density matrices of the steps next to it have rank >= m up to m ~ 500 (its enlarged block has 512 states).  Every m is EVEN: at the centre step
system and environment are the same block, the density-matrix spectra of the sectors q and 1 - q are then identical, every eigenvalue
comes in an exact pair and an odd m would cut one.  Two things differ at this size:
  * the oracle solves superblocks above 3 000 states matrix-free (oracle/kron.py: KronSumOperator, the same operator as the explicit
    matrix, checked in tests/test_oracle_golden.py) -- the explicit sparse matrix of a 10^5-state superblock takes minutes per step;
  * "well-defined cut" is stated in ABSOLUTE terms.  At m ~ 300 the spectrum of a density matrix is dense (relative spacing of
    neighbouring eigenvalues ~ 10 %), so the medium-m rule "relative gap > 1e-2" fails at one cut in ten by chance; what decides
    whether two correct implementations keep the same subspace is the rotation round-off can induce between the last kept and the first
    dropped eigenvector, eps |rho| / (lk - ld), and its effect on later energies, ~ (that angle)^2 (lk - ld): with eps ~ 1e-15 an
    absolute gap above 1e-12 keeps both far below the 1e-10 the comparison asks for.  Both numbers (lk, ld) of every cut are recorded,
    so the test can state the rule it applies.
Run time: ~6 minutes on 4 cores (python3 tests/golden/make_engine_golden_large_m.py); writes tests/golden/engine_large_m.json.
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASE = dict(Lx=4, Ly=8, J1=1.0, Jz1=0.8, J2=0.5, Jz2=0.3, qn_sector=1, min_block=8, mwarmup=6,
            msweeps=[8, 12, 18, 26, 38, 56, 82, 120, 174, 252, 366, 480])
KEYS = ("NSites_Sys", "NSites_Env", "NStates_SysEnl", "NStates_EnvEnl", "NumStates_H", "NStates_SysRot", "NStates_EnvRot", "GSEnergy", "TruncErr_Sys", "TruncErr_Env")
ABS_GAP = 1e-12


def well_defined(s):
    ok = True
    for side in ("Sys", "Env"):
        lk, ld = s["cut_" + side]
        full = s["NStates_%sRot" % side] == s["NStates_%sEnl" % side]
        ok = ok and (full or (lk - ld) > ABS_GAP)
    return bool(ok)


if __name__ == "__main__":
    from oracle.hamiltonian import J1J2XXZModel_SquareLattice
    from oracle.dmrg import DMRGOracle
    c = dict(CASE)
    if len(sys.argv) > 1:
        c["msweeps"] = [int(x) for x in sys.argv[1].split(",")]
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "engine_large_m.json")
    H = J1J2XXZModel_SquareLattice(Lx=c["Lx"], Ly=c["Ly"], J1=c["J1"], Jz1=c["Jz1"], J2=c["J2"], Jz2=c["Jz2"])
    o = DMRGOracle(H, c["mwarmup"], qn_sector=float(c["qn_sector"]), matrix_free_above=3000)
    t0 = time.time()
    o.Warmup()
    m_of_step = [c["mwarmup"]] * len(o.steps)
    for m in c["msweeps"]:
        n0, t = len(o.steps), time.time()
        o.SingleSweep(m, min_block=c["min_block"])
        st = o.steps[n0:]
        m_of_step += [m] * len(st)
        gaps = [min((s["cut_" + sd][0] - s["cut_" + sd][1]) for sd in ("Sys", "Env") if s["NStates_%sRot" % sd] != s["NStates_%sEnl" % sd]) if any(s["NStates_%sRot" % sd] != s["NStates_%sEnl" % sd] for sd in ("Sys", "Env")) else 1.0 for s in st]
        print(f"m {m}: {len(st)} steps in {time.time() - t:.0f} s, ill-defined {[i for i, s in enumerate(st) if not well_defined(s)]}, smallest absolute gap {min(gaps):.2e}, "
              f"largest sector {max(max(s['sectors_SysEnl'][1]) for s in st)}, largest superblock {max(s['NumStates_H'] for s in st)}, E {st[-1]['GSEnergy']:.12f}", flush=True)
    steps = [dict({k: (int(s[k]) if k.startswith("N") else float(s[k])) for k in KEYS}, well_defined=well_defined(s), m=int(m), max_sector=int(max(s["sectors_SysEnl"][1])),
                  cut_Sys=[float(v) for v in s["cut_Sys"]], cut_Env=[float(v) for v in s["cut_Env"]]) for s, m in zip(o.steps, m_of_step)]
    first_ill = next((i for i, s in enumerate(steps) if not s["well_defined"]), len(steps))
    out = {"j1j2_%dx%d_sz1_large_m" % (c["Lx"], c["Ly"]): dict(options={k: c[k] for k in ("Lx", "Ly", "J1", "Jz1", "J2", "Jz2")}, qn_sector=c["qn_sector"], min_block=c["min_block"], mwarmup=c["mwarmup"],
                                        msweeps=c["msweeps"], abs_gap=ABS_GAP, steps=steps, first_ill=first_ill, seconds=time.time() - t0)}
    json.dump(out, open(out_path, "w"), indent=0)
    print("first ill-defined cut at step", first_ill, "of", len(steps), "->", out_path)
