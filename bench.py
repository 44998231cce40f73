#!/usr/bin/env python3
"""Benchmark of the DMRG hot path on MI355X: superblock MatMults/s inside the ground-state eigensolve.

A "step" is one Lanczos step of the eigensolve on one synthetic superblock of a BASELINE.json config:
one superblock MatMult (the Sz-sector block-sparse Kronecker apply, == MatMult_KronSumShell) plus its
reorthogonalisation passes and the amortised thick restart.  Inputs (operators, vectors) are resident in HBM
before the timed region.  value = MatMults/s of the whole job (all ranks work on ONE superblock: strong scaling).

  python bench.py --gpus N --steps K --warmup W [--workload cfg4]
  N > 1: launched by torch.distributed.run, one rank per GPU, RCCL all-gather / all-reduce over xGMI.

The JSON line also carries `roofline` (dominant kernel = the grouped MFMA-f64 GEMM, timed with HIP events on
its own stream inside the timed region) and `cpu_baseline` (the oracle's literal restatement of the
reference's row loop, timed on this host's cores on a bounded row sample; N = 1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
F64_PEAK_TFLOPS = 78.6   # MI355X dense f64 MFMA peak (AMD datasheet; v_mfma_f64_16x16x4 = 64 cycles/SIMD measured:
                         # 72-75 TF/s sustained at the clock the chip holds, profiles/r01_mfma_f64_probe.txt)
HBM_PEAK_TBS = 8.0


def cpu_baseline(sb, budget_s=15.0, gpu_apply=None):
    """Literal reference row loop (oracle/kron_ref.c) on all host cores over evenly spread row chunks."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_shell_from_superblock
    from oracle.kron_c import ShellApplyC
    shell = oracle_shell_from_superblock(sb)
    ref = ShellApplyC(shell)
    N = shell.N
    x = np.random.default_rng(0).standard_normal(N)
    total_flops = float(ref.flops())
    nthreads = int(os.environ.get("DMRGX_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))   # box share: 16 cores per GPU
    nchunk = 8

    y_gpu = gpu_apply(x) if gpu_apply is not None else None
    scale = float(np.abs(y_gpu).max()) if y_gpu is not None else 1.0
    parity = [0.0]

    def run(rows):
        rows = int(min(max(rows, 8 * nthreads), N // nchunk))
        t, f = 0.0, 0.0
        for c in range(nchunk):
            r0 = max(0, min(N - rows, int((c + 0.5) * N / nchunk) - rows // 2))
            t0 = time.perf_counter()
            yr = ref.apply(x, r0, r0 + rows, nthreads)
            t += time.perf_counter() - t0
            f += float(ref.flops(r0, r0 + rows))
            if y_gpu is not None:       # the sampled rows of the literal row loop against the timed GPU path (checker, outside every timed region)
                parity[0] = max(parity[0], float(np.abs(yr[r0:r0 + rows] - y_gpu[r0:r0 + rows]).max() / scale))
        return t, f, rows * nchunk
    rows = 8 * nthreads
    t, f, nrows = run(rows)
    while t < budget_s / 4 and nrows < N // 2:      # grow the sample until it is a meaningful fraction of the budget
        rows = int(rows * min(8.0, max(2.0, 0.8 * budget_s / max(t, 1e-3))))
        t, f, nrows = run(rows)
    full_time = t * total_flops / f
    # the row loop adds the ~1e7 products of a row one after the other: its own rounding error is eps sqrt(products per row) relative
    # to the row's magnitude (the two CPU statements differ by 1.5e-13 at m = 2048) -- the bar of tests/test_gpu_kron.py
    tol_rows = max(1e-13, float(np.finfo(float).eps) * (total_flops / 2.0 / N) ** 0.5)
    if y_gpu is not None:
        assert parity[0] <= tol_rows, f"GPU MatMult differs from the reference row loop on the sampled rows: {parity[0]} > {tol_rows}"
    return {"value": 1.0 / full_time, "unit": "MatMults/s", "cores": int(ref.threads_used), "kind": "port",
            "parity_max_rel_err": parity[0] if y_gpu is not None else None, "parity_tolerance": tol_rows,
            "sample": f"{nrows} of {N} rows in {nchunk} evenly spread chunks ({t:.1f} s measured, "
                      f"{f / t / 1e9:.1f} GF/s unfactored), extrapolated by the row loop's exact flop count "
                      f"({total_flops / 1e9:.0f} GF per MatMult)"}


def cpu_baseline_factored(sb, reps=5, gpu_apply=None):
    """the same MatMult in factored, operator-merged form (oracle/kron_factored.py: per-sector numpy / OpenBLAS GEMMs,
    exactly F_alg flops) on this host's cores, timed IN FULL (every row of the superblock; median of `reps` after one warm-up)."""
    from oracle.kron_factored import FactoredApplyCPU
    nthreads = int(os.environ.get("DMRGX_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=nthreads)
    except Exception:
        limit = None
    f = FactoredApplyCPU(sb)
    x = np.random.default_rng(0).standard_normal(sb.n_states)
    y_cpu = f.apply(x)
    # parity of the timed GPU MatMult with this CPU statement of it, on the same x: the checker, after the
    # timed region.  Same factorisation and blocked sums on both sides: the parity tests' 1e-13 of max|y| holds at this size.
    parity = None
    if gpu_apply is not None:
        y_gpu = gpu_apply(x)
        parity = float(np.abs(y_gpu - y_cpu).max() / np.abs(y_cpu).max())
        assert parity <= 1e-13, f"GPU MatMult differs from the CPU oracle at bench size: {parity}"
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f.apply(x)
        ts.append(time.perf_counter() - t0)
    if limit is not None:
        limit.restore_original_limits()
    t = sorted(ts)[len(ts) // 2]
    return {"value": 1.0 / t, "unit": "MatMults/s", "cores": nthreads, "kind": "port-factored", "parity_max_rel_err": parity,
            "sample": f"all {sb.n_states} rows, {reps} full applies after one warm-up (median {t:.3f} s, {f.flops / t / 1e9:.0f} GF/s on the "
                      f"{f.flops / 1e9:.1f} GF of SURVEY 8d's F_alg: terms merged per right operator, structural zeros of O(x)1 skipped)"}


def engine_run(opts, timeout=300, ranks=1, rehearsal=False):
    """Run the drop-in sweep engine (dmrg.x_amd/dmrgx-square-lattice, the host C++ driver over the same C ABI) in child
    processes and return its DMRGRun.json.  Child processes: they own their HIP contexts, nothing is exec'd from this one.
    ranks > 1: one engine process per GPU with the environment its communicator start-up reads (RANK / WORLD_SIZE / LOCAL_RANK,
    petsc_compat.hpp::CommBootstrap: RCCL id through a rendezvous file); rehearsal: all ranks on GPU 0, host-staged back-end."""
    import subprocess
    import tempfile
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dmrg.x_amd", "dmrgx-square-lattice")
    with tempfile.TemporaryDirectory(prefix="dmrgx_bench_") as d:
        cmd = [exe, *[str(o) for o in opts], "-data_dir", d + "/"]
        if ranks == 1:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}     # a one-rank engine, whatever launched us
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
            if r.returncode != 0:
                raise RuntimeError("sweep engine failed: " + (r.stdout + r.stderr)[-1000:])
        else:
            base = dict(os.environ, WORLD_SIZE=str(ranks), HSA_ENABLE_IPC_MODE_LEGACY="0", DMRGX_RDZV_FILE=os.path.join(d, "rdzv"),
                        DMRGX_SHM_NAME="dmrgx_bench_eng_%d" % os.getpid(), DMRGX_COMM="shm" if rehearsal else "rccl",
                        DMRGX_LAUNCH_NONCE="bench_%d_%d" % (os.getpid(), time.time_ns()))      # the rendezvous file says which launch it belongs to
            # every rank writes its output to a file of its own: a pipe that nobody drains while rank 0 is awaited blocks its writer
            # after 64 KB, the blocked rank stops taking part in the collectives and the whole job hangs
            logs = [open(os.path.join(d, "rank%d.log" % r), "w") for r in range(ranks)]
            procs = [subprocess.Popen(cmd, env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=logs[r], stderr=subprocess.STDOUT) for r in range(ranks)]
            t_end = time.time() + timeout
            rc = 0
            for p in procs:
                try:
                    p.wait(timeout=max(1.0, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    for q in procs:
                        q.kill()                                   # the exact processes started above
                    raise RuntimeError("sweep engine on %d ranks timed out after %d s" % (ranks, timeout))
                rc = rc or p.returncode
            for f in logs:
                f.close()
            if rc != 0:
                raise RuntimeError("sweep engine on %d ranks failed: %s" % (ranks, "".join(open(os.path.join(d, "rank%d.log" % r)).read()[-800:] for r in range(ranks))))
        run = json.load(open(os.path.join(d, "DMRGRun.json")))
        # per-sweep totals from the reference-format step tables (DMRGSteps.json: LoopType/LoopIdx, Timings.json: Total, MatMults)
        steps = json.load(open(os.path.join(d, "DMRGSteps.json")))["table"]
        times = json.load(open(os.path.join(d, "Timings.json")))["table"]
        per = {}
        for st, tm in zip(steps, times):
            if st[1] != "Sweep":
                continue
            e = per.setdefault(int(st[2]), {"steps": 0, "seconds": 0.0, "matmults": 0})
            e["steps"] += 1; e["seconds"] += float(tm[1]); e["matmults"] += int(tm[7])
        run["PerSweep"] = [per[k] for k in sorted(per)]
        # where a step of the last sweep goes on rank 0 (Timings.json: host clock per phase; AllGatherMs / ApplyMs: HIP events on the solver's
        # stream, dmrgx_eigs_comm_timing -- zero on one rank): the N > 1 line must explain itself
        hdr = json.load(open(os.path.join(d, "Timings.json")))["headers"]
        col = {h: i for i, h in enumerate(hdr)}
        last = max(per) if per else None
        rows = [tm for st, tm in zip(steps, times) if st[1] == "Sweep" and int(st[2]) == last]
        if rows:
            mean = lambda name: sum(float(r[col[name]]) for r in rows) / len(rows)          # noqa: E731
            run["StepBreakdownMs"] = {
                "t_step_ms": 1e3 * mean("Total"), "t_solve_ms": 1e3 * mean("Diag"), "t_rdm_ms": 1e3 * mean("Rdms"),
                "t_allgather_ms": mean("AllGatherMs") if "AllGatherMs" in col else None, "t_apply_ms": mean("ApplyMs") if "ApplyMs" in col else None,
                # everything every rank does alike between two solves: enlargement, plan build, start vector, rotation of the coupling operators
                "t_replicated_ms": 1e3 * (mean("Total") - mean("Diag") - mean("Rdms")),
                "rank": 0, "sweep": last, "steps": len(rows)}
        # -step_profile: algorithmic flops and HIP-event GEMM time of every MatMult of every sweep step (KronStats.json)
        ks = [k for k in json.load(open(os.path.join(d, "KronStats.json"))) if k["LoopType"] == "Sweep" and k["timed_applies"] > 0]
        if ks:
            fl = sum(k["flops_alg"] * k["timed_applies"] for k in ks)
            tg = sum(k["ms_stage1"] + k["ms_stage2"] for k in ks) * 1e-3
            run["InSweep"] = {"gemm_tflops": fl / tg / 1e12, "roofline_frac": fl / tg / 1e12 / F64_PEAK_TFLOPS,
                              "mean_flops_alg_per_matmult": fl / sum(k["timed_applies"] for k in ks),
                              "mean_n_states": sum(k["n_states"] for k in ks) / len(ks), "gemm_seconds": tg}
        # what a MatMult of the LAST sweep costs beyond its GEMM launches: (Diag - sum of HIP-event GEMM time) / MatMults --
        # the solver's vector kernels, the projected problem, the split-K fix-up, launch gaps and synchronisations
        if ks and last is not None:
            sweep_of = {int(st[0]): int(st[2]) for st in steps if st[1] == "Sweep"}
            kl = [k for k in ks if sweep_of.get(int(k["GlobIdx"])) == last]
            nmv = sum(int(tm[col["MatMults"]]) for tm in rows) if "MatMults" in col else 0
            if kl and nmv:
                gemm_ms = sum((k["ms_stage1"] + k["ms_stage2"]) * (k["matmults"] / k["timed_applies"]) for k in kl if k["timed_applies"] > 0)
                # the solver's own wall time (entry to final synchronisation; under -step_profile the engine drains the stream in front of it, so
                # the plan's operator copies and the start-vector GEMMs are not in it) -- the `Diag` column also holds those
                solve_ms = 1e3 * sum(k["eigs_seconds"] for k in kl)
                run["NonGemmMsPerMatMult"] = (solve_ms - gemm_ms) / nmv
                run["DiagMinusGemmMsPerMatMult"] = (1e3 * sum(float(r[col["Diag"]]) for r in rows) - gemm_ms) / nmv
                run["GemmMsPerMatMult"] = gemm_ms / nmv
        run["SolverPath"] = {k: run.get(k) for k in ("RdmCalls", "RdmBlockJacobiCalls", "TridPersistentCalls", "TridLaunchPathCalls", "TridFallbacks",
                                                     "TridMaxWorkgroupsPerMatrix", "RdmMaxMergeLevels", "RdmMaxWyBlocks")}
        run["SweepEnergies"] = {}
        for st in steps:
            if st[1] == "Sweep":
                run["SweepEnergies"][int(st[2])] = float(st[-1])          # energy of the last step of every sweep
        run["MaxTruncErr"] = max(float(st[-3]) for st in steps)
        return run


def sweep_legs():
    """The other two parts of BASELINE.json's metric, measured on the real engine (real Hamiltonian, real sweeps):
    sites/sec per sweep on configs[3] -- the lattice and m the north star quotes (J1-J2 20x8 cylinder, J2 = 0.5, m = 2048: it
    fits one MI355X, about a minute) -- and on configs[1] (J1-J2 8x4, m = 512), and the E0 relative error on configs[0]
    (Heisenberg 16x1 chain, m = 64, 2 sweeps) against exact diagonalisation."""
    def leg(run, config, expect_launch_path=False):
        out = {"sites_per_s": run["LastSweepSteps"] / run["LastSweepSeconds"], "config": config, "sweep_steps": run["LastSweepSteps"],
               "sweep_seconds": run["LastSweepSeconds"], "sweep_matmults": run["LastSweepMatMults"],
               "matmults_per_s_in_sweep": run["LastSweepMatMults"] / run["LastSweepSeconds"], "gs_energy": run["GSEnergy"],
               "max_trunc_err": run["MaxTruncErr"]}
        if "InSweep" in run:       # the roofline of the dominant kernel on the REAL superblocks of the sweep
            out["roofline_frac"] = run["InSweep"]["roofline_frac"]
            out["in_sweep"] = run["InSweep"]
        if "NonGemmMsPerMatMult" in run:
            out["nongemm_ms_per_matmult"] = run["NonGemmMsPerMatMult"]          # (solver wall time - GEMM events) / MatMults
            out["diag_minus_gemm_ms_per_matmult"] = run["DiagMinusGemmMsPerMatMult"]      # the same from the Diag phase (+ start vector, + queued plan copies): round 4's 0.71
            out["gemm_ms_per_matmult"] = run["GemmMsPerMatMult"]
        # which path the density-matrix solver took (dmrgx_rdm_info -> DMRGRun.json): a leg that silently fell back to the slower
        # tridiagonalisation (a bounded-spin time-out latches for the rest of the process) or to the block-Jacobi solver is not a measurement
        # of the code this line describes -- it fails instead of looking 5 ms per step slower
        sp = run["SolverPath"]
        out["solver_path"] = sp
        assert sp["TridFallbacks"] == 0, ("the persistent tridiagonalisation timed out and fell back to one launch per column", config, sp)
        if not expect_launch_path:
            assert sp["RdmBlockJacobiCalls"] == 0 and sp["TridLaunchPathCalls"] == 0 and sp["TridPersistentCalls"] == sp["RdmCalls"], ("unexpected density-matrix solver path", config, sp)
        # self-consistency of the printed energies (the parity tier compares them with the oracle, tests/test_gpu_engine.py):
        # DMRG is variational, so the energy at the end of a sweep may not rise above the previous sweep's by more than the
        # truncation error allows
        e = [run["SweepEnergies"][k] for k in sorted(run["SweepEnergies"])]
        tol = 10.0 * max(run["MaxTruncErr"], 1e-12) * abs(e[-1])
        assert all(b <= a + tol for a, b in zip(e, e[1:])), ("sweep energies rise", e)
        assert abs(run["GSEnergy"] - e[-1]) <= 1e-10 * abs(e[-1])          # DMRGSteps.json prints 12 significant digits
        out["sweep_energies"] = e
        return out
    t_legs = time.perf_counter()
    j1j2 = ["-J1", 1, "-Jz1", 1, "-J2", 0.5, "-Jz2", 0.5]
    # -H_eps_type gd: the generalized-Davidson option of the superblock solve (SLEPc users of the reference have the same
    # option name); same convergence criterion, about a quarter fewer MatMults per step than the default Krylov-Schur / Lanczos
    # type from the engine's transformed start vectors (same box: 11.7 vs 10.0 sites/s)
    run3 = engine_run(["-Lx", 20, "-Ly", 8, "-mwarmup", 2048, *j1j2, "-nsweeps", 2, "-step_profile", 1, "-H_eps_type", "gd"], timeout=1200)
    out = leg(run3, "configs[3] on one GPU: J1-J2 20x8 cylinder (160 sites), J2=0.5, m=2048, warm-up + two finite-system sweeps (real engine "
                    "run, -H_eps_type gd); sites_per_s is the second sweep, the first one (environment blocks still from the warm-up) is listed beside it")
    out["per_sweep"] = [{"sites_per_s": p["steps"] / p["seconds"], "matmults_per_s": p["matmults"] / p["seconds"], **p} for p in run3["PerSweep"]]
    # the same lattice is tied to the CPU oracle step by step at m = 6 (tests/test_gpu_engine.py::test_headline_lattices_..., golden
    # table tests/golden/engine_big_lattices.json: E = -121.10624750034 in the Sz = 1 sector); DMRG is variational in m and the
    # Sz = 0 ground state lies below the Sz = 1 one, so the m = 2048 energy must lie below it
    assert run3["GSEnergy"] < -121.10624750033969, run3["GSEnergy"]
    # configs[1]: three sweeps (0.3 s): `sites_per_s` stays the FIRST sweep after the warm-up, as in every earlier round (its environment blocks still
    # come from the warm-up: start vectors through basis overlaps, a third more MatMults); the settled rate of the later sweeps is printed beside it
    # -H_eps_type gd here too since round 5 (as on every other leg): with the projected problem on the device it is the faster type at m = 512 as
    # well (same box, first sweep: 304-311 against 294 sites/s, profiles/r05_configs1_solver_type_ab.txt); the default type's first sweep is listed beside it
    run1c = engine_run(["-Lx", 8, "-Ly", 4, "-mwarmup", 512, *j1j2, "-nsweeps", 3, "-H_eps_type", "gd"])
    out["configs_1"] = leg(run1c, "configs[1]: J1-J2 8x4 cylinder, J2=0.5, m=512, warm-up + three finite-system sweeps (real engine run, -H_eps_type gd); sites_per_s is the "
                                  "first sweep, sites_per_s_settled the third, sites_per_s_default_solver the first sweep with the default (Krylov-Schur / Lanczos) type")
    p0, p2 = run1c["PerSweep"][0], run1c["PerSweep"][-1]
    out["configs_1"].update({"sites_per_s": p0["steps"] / p0["seconds"], "sweep_steps": p0["steps"], "sweep_seconds": p0["seconds"], "sweep_matmults": p0["matmults"],
                             "matmults_per_s_in_sweep": p0["matmults"] / p0["seconds"], "sites_per_s_settled": p2["steps"] / p2["seconds"],
                             "sweep_matmults_settled": p2["matmults"]})
    run1d = engine_run(["-Lx", 8, "-Ly", 4, "-mwarmup", 512, *j1j2, "-nsweeps", 1])
    out["configs_1"]["sites_per_s_default_solver"] = run1d["LastSweepSteps"] / run1d["LastSweepSeconds"]
    out["configs_1"]["sweep_matmults_default_solver"] = run1d["LastSweepMatMults"]
    # configs[2]'s lattice and m on ONE GPU (BASELINE names it for 2 GPUs: the N = 2 bench line runs it there); its geometry is tied
    # to the oracle step by step at reduced m (tests/test_gpu_engine.py, golden table: E = -51.658556812379 at m = 4, Sz = 2)
    run2 = engine_run(["-Lx", 16, "-Ly", 6, "-heisenberg", 1, "-mwarmup", 1024, "-nsweeps", 1, "-H_eps_type", "gd"], timeout=600)
    out["configs_2"] = leg(run2, "configs[2] on one GPU: Heisenberg 16x6 cylinder (96 sites), m=1024, one finite-system sweep after warm-up (real engine run, -H_eps_type gd)")
    assert run2["GSEnergy"] < -51.65855681237887, run2["GSEnergy"]
    # The like-for-like figure beside the tuned one: the reference's own solver settings for this path -- Krylov-Schur type (here:
    # thick-restart Lanczos) from a RANDOM start vector, as include/DMRGBlockContainer.hpp:1488-1499 runs every eigensolve
    # (-wavefunction_guess 0).  configs[1] always; configs[3] (warm-up + one sweep, about a minute) unless the legs above already
    # took unusually long on this box or DMRGX_BENCH_REFSETTINGS=0.
    ref1 = engine_run(["-Lx", 8, "-Ly", 4, "-mwarmup", 512, *j1j2, "-nsweeps", 1, "-wavefunction_guess", 0])
    out["configs_1"]["sites_per_s_reference_settings"] = ref1["LastSweepSteps"] / ref1["LastSweepSeconds"]
    out["configs_1"]["sweep_matmults_reference_settings"] = ref1["LastSweepMatMults"]
    out["sites_per_s_reference_settings"] = None
    if os.environ.get("DMRGX_BENCH_REFSETTINGS", "1") != "0" and time.perf_counter() - t_legs < 110.0:
        try:
            ref3 = engine_run(["-Lx", 20, "-Ly", 8, "-mwarmup", 2048, *j1j2, "-nsweeps", 1, "-wavefunction_guess", 0], timeout=900)
            out["sites_per_s_reference_settings"] = ref3["LastSweepSteps"] / ref3["LastSweepSeconds"]
            out["reference_settings"] = {"config": "configs[3], krylovschur-type solver, random start vector (-wavefunction_guess 0), first sweep after the warm-up",
                                         "sweep_matmults": ref3["LastSweepMatMults"], "sweep_seconds": ref3["LastSweepSeconds"], "gs_energy": ref3["GSEnergy"]}
        except Exception as e:                                    # noqa: BLE001 -- reported in the JSON line
            out["reference_settings"] = {"error": str(e)[-400:]}
    e_ed = -6.9117371455751
    run1 = engine_run(["-Lx", 16, "-Ly", 1, "-heisenberg", 1, "-mwarmup", 64, "-nsweeps", 2, "-H_eps_tol", 1e-12])
    out["e0_rel_err"] = abs(run1["GSEnergy"] - e_ed) / abs(e_ed)
    assert out["e0_rel_err"] <= 1e-10, out["e0_rel_err"]                     # north-star tolerance
    # configs[1]: the energy the parity tier ties to the oracle (tests/test_gpu_engine.py::test_baseline_config1_energy_...)
    assert abs(out["configs_1"]["gs_energy"] - (-27.927342512)) <= 1e-6, out["configs_1"]["gs_energy"]
    out["e0_config"] = "configs[0]: Heisenberg 16x1 chain, m=64, 2 sweeps; exact-diagonalisation E0 = -6.9117371455751"
    # configs[4]: XY 32x8 cylinder at m = 4096 on ONE GPU (93 GB peak; the NNN terms drop out with Jz2 = 0 exactly as in the reference,
    # SURVEY section 5).  About two minutes (warm-up + one sweep): run unless the legs above already took unusually long on this
    # box or DMRGX_BENCH_CONFIGS4=0; a failure here is reported, it does not take the bench line down.
    if os.environ.get("DMRGX_BENCH_CONFIGS4", "1") != "0" and time.perf_counter() - t_legs < 200.0:
        try:
            run4 = engine_run(["-Lx", 32, "-Ly", 8, "-J1", 1, "-Jz1", 0, "-J2", 1, "-Jz2", 0, "-mwarmup", 4096, "-nsweeps", 1, "-H_eps_type", "gd"], timeout=900)
            out["configs_4"] = leg(run4, "configs[4] on one GPU: XY 32x8 cylinder (256 sites), m=4096, warm-up + one finite-system sweep (real engine run, -H_eps_type gd)",
                                   expect_launch_path=True)      # (sectors above order ~1700 take one launch per column by design)
            out["configs_4"]["device_bytes_peak"] = run4.get("DeviceBytesPeak")
            out["configs_4"]["device_bytes_resident_after_sweep"] = run4.get("DeviceBytesResidentAfterSweep")
        except Exception as e:                                    # noqa: BLE001 -- reported in the JSON line
            out["configs_4"] = {"error": str(e)[-400:]}
    else:
        out["configs_4"] = None
    return out


def sweep_legs_multi(ranks, rehearsal):
    """N > 1: the configs[3] engine leg on N ranks (same binary, one process per GPU, striped plans, RDMs dealt over the ranks), and
    for N = 2 BASELINE configs[2] as specified (Heisenberg 16x6 cylinder, m = 1024, Sz sectors sharded over 2 GPUs).  The rehearsal
    (tests, one GPU) runs the same control flow on small lattices."""
    def leg(run, config):
        return {"sites_per_s": run["LastSweepSteps"] / run["LastSweepSeconds"], "config": config, "ranks": run.get("Ranks"),
                "sweep_steps": run["LastSweepSteps"], "sweep_seconds": run["LastSweepSeconds"], "sweep_matmults": run["LastSweepMatMults"],
                "matmults_per_s_in_sweep": run["LastSweepMatMults"] / run["LastSweepSeconds"], "gs_energy": run["GSEnergy"], "max_trunc_err": run["MaxTruncErr"],
                "step_breakdown_ms": run.get("StepBreakdownMs")}
    j1j2 = ["-J1", 1, "-Jz1", 1, "-J2", 0.5, "-Jz2", 0.5]
    if rehearsal:
        big = ["-Lx", 6, "-Ly", 4, "-mwarmup", 48, *j1j2, "-nsweeps", 1]
        heis = ["-Lx", 6, "-Ly", 2, "-heisenberg", 1, "-mwarmup", 32, "-nsweeps", 1]
        big_name, heis_name = "rehearsal: J1-J2 6x4, m=48", "rehearsal: Heisenberg 6x2, m=32"
    else:
        big = ["-Lx", 20, "-Ly", 8, "-mwarmup", 2048, *j1j2, "-nsweeps", 2, "-H_eps_type", "gd"]
        heis = ["-Lx", 16, "-Ly", 6, "-heisenberg", 1, "-mwarmup", 1024, "-nsweeps", 2, "-H_eps_type", "gd"]
        big_name = "configs[3] on %d GPUs: J1-J2 20x8 cylinder, J2=0.5, m=2048, warm-up + two sweeps (-H_eps_type gd); sites_per_s is the second sweep" % ranks
        heis_name = "configs[2]: Heisenberg 16x6 cylinder, m=1024, on 2 GPUs, warm-up + two sweeps (-H_eps_type gd)"
    out = {}
    try:
        out = leg(engine_run(big, timeout=1200, ranks=ranks, rehearsal=rehearsal), big_name)
    except Exception as e:                                        # noqa: BLE001 -- reported in the JSON line, the MatMult line stands
        out = {"error": str(e)[-600:]}
    if ranks == 2:
        try:
            out["configs_2"] = leg(engine_run(heis, timeout=900, ranks=2, rehearsal=rehearsal), heis_name)
        except Exception as e:                                    # noqa: BLE001
            out["configs_2"] = {"error": str(e)[-600:]}
    return out


def kernel_source_hash():
    """Identifies the build a PMC measurement belongs to: the sources of the dominant kernel and of the plan that feeds it."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(os.path.abspath(__file__))
    for f in ("dmrg.x_amd/csrc/ggemm.hip", "dmrg.x_amd/csrc/ggemm.h", "dmrg.x_amd/csrc/kron_plan.hip"):
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def launch_ranks(n):
    """`python bench.py --gpus N` as a plain command: start the N rank processes ourselves (what torch.distributed.run would
    do: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), relay rank 0's JSON line, exit with the worst return code.
    Called before anything in this process has touched the GPU; the children are ordinary child processes (no exec of this one)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # DMRGX_BENCH_LAUNCHER=1: the engine legs are run by THIS process once every rank has exited and released its GPU (a rank 0 that
    # starts them itself cannot know that its siblings are gone: ADVICE round 3)
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", DMRGX_BENCH_LAUNCHER="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=dict(base, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL) for r in range(n)]
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    text = out0.decode()
    if rc == 0 and "--no-sweep" not in sys.argv:
        lines = text.strip().splitlines()
        try:
            line = json.loads(lines[-1])
            line["sweep"] = sweep_legs_multi(n, os.environ.get("DMRGX_BENCH_REHEARSAL") == "1")
            text = "\n".join(lines[:-1] + [json.dumps(line)]) + "\n"
        except Exception as e:                                            # noqa: BLE001
            sys.stderr.write("bench launcher: engine legs failed: %s\n" % e)
            rc = 1
    sys.stdout.write(text)
    sys.stdout.flush()
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--workload", default="cfg4real", help="cfg4real (default: BASELINE configs[3], J1-J2 20x8, m=2048, on the sector tables of a real "
                    "mid-sweep step of the engine) | cfg1..cfg5 (SURVEY 8d's synthetic sigma=1.8 sector profile)")
    ap.add_argument("--ncv", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the engine legs (sites/sec per sweep, E0 rel-err)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL; must be in place before the first HIP call
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)                  # plain `python bench.py --gpus N`: become the launcher (never returns)
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE)")
    # Rehearsal mode (tests only): DMRGX_BENCH_REHEARSAL=1 runs the N > 1 control flow with all ranks on cuda:0 and the
    # collectives staged through gloo on the host -- RCCL needs one GPU per rank, the one-GPU test box has one.
    rehearsal = os.environ.get("DMRGX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")          # control plane only (id exchange, barrier, max of the wall times): every
                                                 # data-path collective is RCCL, issued by libdmrgx_hip.so itself

    from __graft_entry__ import load_package
    load_package()
    from dmrgx_amd.superblock import KronPlan, Communicator
    from dmrgx_amd.workloads import synthetic_superblock, CONFIGS

    sb = synthetic_superblock(args.workload)
    plan = KronPlan(sb, device=f"cuda:{local_rank}", world_size=world, rank=rank)
    info = plan.info
    comm = None
    if world > 1:
        # native communicator (csrc/comm.hip): RCCL all-gather / all-reduce enqueued by the eigensolver itself -- no Python in
        # the step loop.  Rank 0's 128-byte RCCL id reaches the other ranks through the launcher's store.
        if rehearsal:
            comm = Communicator(rank, world, host_staged_name="dmrgx_bench_%s" % os.environ.get("MASTER_PORT", "0"))
        else:
            ids = [Communicator.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            comm = Communicator(rank, world, unique_id=ids[0])
    hooks = {"comm": comm} if comm is not None else {}

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    plan.eigs_lowest(ncv=args.ncv, tol=1e-300, seed=1, max_matvec=max(args.warmup, 1), **hooks)   # untimed warm-up
    barrier()
    plan.timing(True)
    t0 = time.perf_counter()
    _, _, stats = plan.eigs_lowest(ncv=args.ncv, tol=1e-300, seed=2, max_matvec=args.steps, **hooks)
    barrier()
    elapsed = time.perf_counter() - t0
    plan.timing(False)
    assert stats.n_matvec == args.steps, (stats.n_matvec, args.steps)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms4, napp = plan.timing_read()
    ms1, ms2 = ms4[0] + ms4[1], ms4[2] + ms4[3]
    # dominant kernel: the grouped MFMA-f64 GEMM; 64x64 instantiation unless DMRGX_TILES=mixed routes the
    # 128-aligned cores to the 128x128 one.  Two launches of it per MatMult (stage 1, stage 2).
    launches = 2 * napp
    if info.n_tiles_big > 0:
        kernel, avg_ms, flops_per_launch = "dmrgx::ggemm_kernel_128", (ms4[0] + ms4[2]) / max(launches, 1), info.flops_alg_big / 2.0
    else:
        kernel, avg_ms, flops_per_launch = "dmrgx::ggemm_kernel_64", (ms4[1] + ms4[3]) / max(launches, 1), info.flops_alg / 2.0
    achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    gemm_all = info.flops_alg / max((ms1 + ms2) / max(napp, 1) * 1e-3, 1e-12) / 1e12     # all four GEMM launches together

    # K1 in isolation (MatMult only, no Lanczos vector work)
    x = torch.randn(info.vec_len, dtype=torch.float64, device="cuda")
    y = torch.zeros(info.vec_len, dtype=torch.float64, device="cuda")
    yl = y[info.local_offset:info.local_offset + info.local_len]
    for _ in range(3):
        plan.apply(x, yl)
    barrier()
    t1 = time.perf_counter()
    for _ in range(20):
        plan.apply(x, yl)
    barrier()
    iso = 20 / (time.perf_counter() - t1)

    out = {
        "metric": "superblock MatMults/sec inside the ground-state eigensolve (Lanczos step = 1 MatMult + reorthogonalisation)",
        "value": args.steps / elapsed, "unit": "MatMults/s", "n_gpus": comm.info()[1] if comm is not None else 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {CONFIGS[args.workload]['desc']}" + ("" if args.workload == "cfg4real" else "; mid-sweep column cut, sector profile sigma=1.8 (SURVEY 8d)"),
                   "m": CONFIGS[args.workload]["m"], "n_states": int(info.n_states), "n_terms": len(sb.terms) + 2,
                   "ncv": args.ncv, "parallelism": f"right-index stripes over {world} GPU(s)"},
        "matmult_isolated_per_s": iso,
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / F64_PEAK_TFLOPS, "traffic": None,
                     "kernel": kernel, "avg_launch_ms": avg_ms, "launches": launches,
                     "flops_alg_per_launch": flops_per_launch, "all_gemm_launches_tflops": gemm_all,
                     "ms_per_matmult_by_launch": {"stage1_128": ms4[0] / max(napp, 1), "stage1_64": ms4[1] / max(napp, 1),
                                                  "stage2_128": ms4[2] / max(napp, 1), "stage2_64": ms4[3] / max(napp, 1)},
                     "tiles_128": info.n_tiles_big,
                     "flops_alg_per_matmult": info.flops_alg, "bytes_alg_per_matmult": info.bytes_alg,
                     "flops_exec_per_matmult": info.flops_exec,
                     "stage1_ms_per_matmult": ms1 / max(napp, 1), "stage2_ms_per_matmult": ms2 / max(napp, 1),
                     "tiles_stage1": info.n_tiles_stage1, "tiles_stage2": info.n_tiles_stage2,
                     "hbm_frac_of_peak": (info.bytes_alg + info.bytes_workspace) / max((ms1 + ms2) / max(napp, 1) * 1e-3, 1e-12) / (HBM_PEAK_TBS * 1e12)},
    }
    # PMC-measured fabric traffic of the same command (rocprofv3 --pmc passes, tools/profile.sh -> tools/make_traffic_json.py).  It
    # is a committed measurement, not something this run measures: it is only quoted if it was taken on THIS kernel and plan
    # builder (hash of their sources recorded beside it), and the line says at which commit it was measured.
    traffic_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        t = json.load(open(traffic_file)).get(f"{args.workload}@{world}")
        if t and t.get("kernel") == kernel:
            if t.get("source_hash") == kernel_source_hash():
                out["roofline"]["traffic"] = t["bytes_per_launch"]
                out["roofline"]["traffic_source"] = t["source"]
                out["roofline"]["traffic_measured_at_commit"] = t.get("measured_at_commit")
            else:
                out["roofline"]["traffic_source"] = ("profiles/traffic.json was measured on other kernel / plan sources (hash %s, running %s): not quoted; "
                                                     "regenerate with tools/profile.sh + tools/make_traffic_json.py" % (t.get("source_hash"), kernel_source_hash()))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # two CPU statements of the same MatMult on this host: the factored, operator-merged form -- the same
        # algorithm as the HIP plan, so value / cpu_baseline.value is hardware against hardware -- and beside it the literal
        # unfactored row loop the reference executes (sampled rows, extrapolated by its exact flop count)
        def gpu_apply(x_host):
            xd = torch.from_numpy(x_host).cuda()
            yd = torch.full_like(xd, float("nan"))
            plan.apply(xd, yd)
            torch.cuda.synchronize()
            return yd.cpu().numpy()
        out["cpu_baseline"] = cpu_baseline_factored(sb, gpu_apply=gpu_apply)
        out["parity_max_rel_err"] = out["cpu_baseline"]["parity_max_rel_err"]      # |y_gpu - y_oracle|_max / |y_oracle|_max, full-size MatMult
        out["cpu_baseline"]["reference_row_loop"] = cpu_baseline(sb, gpu_apply=gpu_apply)
        out["parity_rows_max_rel_err"] = out["cpu_baseline"]["reference_row_loop"]["parity_max_rel_err"]
    elif rank == 0:
        out["cpu_baseline"] = None
    if comm is not None:
        out["communicator"] = dict(zip(("rank", "world", "backend"), comm.info()), backend_names={"0": "rccl", "1": "host-staged (rehearsal)"})
    plan.destroy()
    if comm is not None:
        comm.destroy()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    torch.cuda.empty_cache()                     # the engine legs below run in child processes on the same GPU(s)
    if rank == 0 and not args.no_sweep and not (world > 1 and os.environ.get("DMRGX_BENCH_LAUNCHER") == "1"):
        # sites/sec per sweep on the real engine: one rank -> the full set of legs; N ranks -> the configs[3] engine leg on N ranks
        # (and configs[2] for N = 2).  Started by `python bench.py --gpus N` the launcher process runs the N-rank legs after all ranks
        # have exited; under torch.distributed.run rank 0 does, after giving its siblings a moment to leave their GPUs.
        if world > 1:
            time.sleep(3.0)
        out["sweep"] = sweep_legs() if world == 1 else sweep_legs_multi(world, rehearsal)
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
