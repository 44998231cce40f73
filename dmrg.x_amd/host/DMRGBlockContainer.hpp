/** @file DMRGBlockContainer.hpp
    DMRGBlockContainer<Block, Hamiltonian>: warm-up, sweeps and the single DMRG step, with the public interface, the
    option names, the block-index schedule and the truncation rules of the reference orchestrator (reference
    include/DMRGBlockContainer.hpp:166-2777) so that src/DMRG-SquareLattice.cpp compiles on top of it unchanged.
    What a step does here:

        enlarge   KronEye_Explicit            cell views + device H assembly              (host metadata, HBM data)
        build H   KronBlocks.KronSumConstruct dmrgx_kron_plan_create                      (matrix-free, MFMA GEMMs)
        solve     dmrgx_eigs_lowest           thick-restart Lanczos, psi stays in HBM     (replaces SLEPc EPSSolve)
        truncate  GetTruncation               dmrgx_rdm_create (RDMs + block Jacobi), host sort / m-cut / sector re-sort
        rotate    Block::RotateOperators      dmrgx_rotate_ops, all operators in one call

    The eigensolver is configured with the reference's option prefix: -H_eps_tol, -H_eps_ncv, -H_eps_max_it.
    DMRGSteps.json / Timings.json keep the reference's tabular schema (Timings.json gains a MatMults column). */
#ifndef DMRGX_DMRGBLOCKCONTAINER_HPP
#define DMRGX_DMRGBLOCKCONTAINER_HPP

#include <algorithm>
#include <cmath>
#include <climits>
#include <iomanip>
#include <fstream>
#include <array>
#include <sstream>
#include <iostream>
#include <map>
#include <set>
#include <string>
#include <vector>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include "DMRGKron.hpp"
#include "CorrelatorDealing.hpp"

/** One eigenpair of a reduced-density-matrix block */
struct Eigen_t
{
    PetscScalar eigval;  /**< eigenvalue */
    PetscInt    seqIdx;  /**< KronBlock the matrix belongs to */
    PetscInt    epsIdx;  /**< rank inside that block's spectrum (0 = largest) */
    PetscInt    blkIdx;  /**< sector index in the block's Magnetization */
};
inline bool greater_eigval(const Eigen_t& a, const Eigen_t& b) { return a.eigval > b.eigval; }
inline bool less_blkIdx(const Eigen_t& a, const Eigen_t& b) { return a.blkIdx < b.blkIdx; }

/** An operator of a measurement */
struct Op {
    Op_t     OpType;
    PetscInt idx;
    PetscErrorCode PrintInfo() const { std::cout << "  Op" << OpToStr(OpType) << idx << std::endl; return 0; }
};

/** Formats and writes records off the sweep's critical path: a step's entanglement spectra are ~8 k numbers (1 ms of fprintf
    at m = 2048, with the GPU idle behind it); the step hands them over and goes on.  One worker, jobs run in submission order;
    Drain() returns when everything handed over so far is in the file. */
class BackgroundWriter {
public:
    void Push(std::function<void()> job)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (!worker.joinable()) worker = std::thread([this] { Run(); });
        jobs.push_back(std::move(job));
        cv.notify_one();
    }
    void Drain()
    {
        std::unique_lock<std::mutex> lk(mu);
        idle.wait(lk, [this] { return jobs.empty() && !busy; });
    }
    ~BackgroundWriter()
    {
        { std::unique_lock<std::mutex> lk(mu); stop = true; cv.notify_one(); }
        if (worker.joinable()) worker.join();
    }
private:
    void Run()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [this] { return stop || !jobs.empty(); });
            if (jobs.empty()) return;                       /* stop requested and nothing left */
            std::function<void()> job = std::move(jobs.front());
            jobs.pop_front();
            busy = true;
            lk.unlock();
            job();
            lk.lock();
            busy = false;
            if (jobs.empty()) idle.notify_all();
        }
    }
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv, idle;
    std::deque<std::function<void()>> jobs;
    bool stop = false, busy = false;
};

template<class Block, class Hamiltonian> class DMRGBlockContainer
{
public:
    /** Result of the truncation of one side */
    struct BasisTransformation
    {
        Mat RotMatT;            /**< rotation matrix (sector-block rows of kept eigenvectors) */
        QuantumNumbers QN;      /**< sectors of the truncated block */
        PetscReal TruncErr = 0; /**< 1 - sum of kept (positive) eigenvalues */
    };

    /** one (old sector, site state) piece of a sector of `block (x) site` */
    struct EnlPart { int32_t old_sector, site_sector, off, size; };
    /* (see block_rot below) */
    struct RotMeta {
        int64_t parent_ver = -1;                          /**< version of sys_blocks[i-1] the rotation was computed in */
        std::vector<std::vector<EnlPart>> parts;          /**< enlarged sectors of (that parent (x) site) */
        std::vector<int> parent_qn2, enl_qn2;             /**< 2 Sz of the parent's sectors / of the enlarged sectors */
        std::vector<int32_t> enl_sizes;
    };
    struct BasisOverlap {                                 /**< <version from | version to> of one block, sector by sector (key 2 Sz) */
        int64_t from_ver = -1, to_ver = -1;
        struct Cell { int32_t rows = 0, cols = 0; std::shared_ptr<dmrgx_host::DevBuffer> buf; };
        std::map<int, Cell> cells;
    };

    explicit DMRGBlockContainer(const MPI_Comm& mpi_comm) : mpi_comm(mpi_comm) {}
    ~DMRGBlockContainer() { PetscErrorCode ierr = Destroy(); CPP_CHKERR(ierr); }

    PetscErrorCode Initialize()
    {
        if (init) SETERRQ(mpi_comm, 1, "DMRG object has already been initialized.");
        PetscErrorCode ierr;
        char path[PETSC_MAX_PATH_LEN];
        ierr = MPI_Comm_size(mpi_comm, &mpi_size); CHKERRQ(ierr);
        ierr = MPI_Comm_rank(mpi_comm, &mpi_rank); CHKERRQ(ierr);
        /*  checkpoint restart (include/DMRGBlockContainer.hpp:282-361 of the reference): -restart_dir points to the scratch
            directory of a previous run; the last Sweep_%09d with a Sweep.dat is resumed, its Hamiltonian.dat overrides
            the model options and, with -restart_options, its PetscOptions.dat the sweep schedule */
        ierr = PetscOptionsGetString(NULL, NULL, "-restart_dir", path, PETSC_MAX_PATH_LEN, &restart); CHKERRQ(ierr);
        ierr = PetscOptionsGetBool(NULL, NULL, "-restart_options", &restart_options, NULL); CHKERRQ(ierr);
        if (restart) {
            restart_dir = std::string(path);
            if (restart_dir.back() != '/') restart_dir += '/';
            const PetscInt MAX_SWEEP_IDX = 1000000;
            PetscInt ridx = 0;
            PetscBool flg = PETSC_FALSE;
            while (ridx < MAX_SWEEP_IDX) { ierr = PetscTestDirectory((restart_dir + SweepDir(ridx)).c_str(), 'r', &flg); CHKERRQ(ierr); if (flg) break; ++ridx; }
            if (ridx == MAX_SWEEP_IDX) SETERRQ1(mpi_comm, 1, "No Sweep directory was found in %s", restart_dir.c_str());
            while (ridx < MAX_SWEEP_IDX) { ierr = PetscTestFile((restart_dir + SweepDir(ridx + 1) + "Sweep.dat").c_str(), 'r', &flg); CHKERRQ(ierr); if (flg) ++ridx; else break; }
            restart_dir += SweepDir(ridx);
            ierr = SetOptionsFromFile(restart_dir + "Hamiltonian.dat"); CHKERRQ(ierr);
            std::map<std::string, PetscInt> dict;
            ierr = RetrieveInfoFile(restart_dir + "Sweep.dat", dict); CHKERRQ(ierr);
            for (const char* k : {"GlobIdx", "LoopIdx", "num_sys_blocks", "sys_ninit", "num_sites"})
                if (!dict.count(k)) SETERRQ2(mpi_comm, 1, "%sSweep.dat: key %s missing.", restart_dir.c_str(), k);
            GlobIdx = dict["GlobIdx"]; LoopIdx = dict["LoopIdx"] + 1;      /* saved before the increment */
            num_sys_blocks = dict["num_sys_blocks"]; restart_sys_ninit = dict["sys_ninit"]; restart_num_sites = dict["num_sites"];
            if (restart_options) {
                ierr = SetOptionsFromFile(restart_dir + "PetscOptions.dat"); CHKERRQ(ierr);
                if (dict.count("msweep_idx")) restart_msweep_idx = dict["msweep_idx"];
            }
            if (!mpi_rank) printf("RESTART from %s  (GlobIdx %lld, LoopIdx %lld, %lld blocks)\n", restart_dir.c_str(), LLD(GlobIdx), LLD(LoopIdx), LLD(restart_sys_ninit));
        }
        ierr = Ham.SetFromOptions(); CHKERRQ(ierr);
        ierr = SingleSite.Initialize(mpi_comm, 1, PETSC_DEFAULT); CHKERRQ(ierr);
        num_sites = Ham.NumSites();
        if (num_sites < 2) SETERRQ1(mpi_comm, 1, "There must be at least two total sites. Got %lld.", LLD(num_sites));
        if (num_sites % 2) SETERRQ1(mpi_comm, 1, "Total number of sites must be even. Got %lld.", LLD(num_sites));
        ierr = PetscOptionsGetBool(NULL, NULL, "-verbose", &verbose, NULL); CHKERRQ(ierr);
        ierr = PetscOptionsGetBool(NULL, NULL, "-no_symm", &no_symm, NULL); CHKERRQ(ierr);
        ierr = PetscOptionsGetBool(NULL, NULL, "-do_shell", &do_shell, NULL); CHKERRQ(ierr);
        ierr = PetscOptionsGetBool(NULL, NULL, "-dry_run", &dry_run, NULL); CHKERRQ(ierr);
        ierr = PetscOptionsGetReal(NULL, NULL, "-qn_sector", &qn_sector, NULL); CHKERRQ(ierr);
        ierr = PetscOptionsGetReal(NULL, NULL, "-H_eps_tol", &eps_tol, NULL); CHKERRQ(ierr);
        ierr = PetscOptionsGetInt(NULL, NULL, "-H_eps_ncv", &eps_ncv, NULL); CHKERRQ(ierr);
        ierr = PetscOptionsGetInt(NULL, NULL, "-H_eps_gd_minv", &eps_gd_minv, NULL); CHKERRQ(ierr);
        ierr = PetscOptionsGetInt(NULL, NULL, "-H_eps_max_it", &eps_max_it, NULL); CHKERRQ(ierr);
        {   /* SLEPc's solver choice for the superblock problem: krylovschur (its default; here thick-restart Lanczos) or gd */
            char type[64] = "krylovschur"; PetscBool set = PETSC_FALSE;
            ierr = PetscOptionsGetString(NULL, NULL, "-H_eps_type", type, sizeof(type), &set); CHKERRQ(ierr);
            const std::string t(type);
            if (t == "gd") eps_method = 1;
            else if (t == "krylovschur" || t == "lanczos") eps_method = 0;
            else SETERRQ1(mpi_comm, PETSC_ERR_SUP, "-H_eps_type %s is not available (krylovschur, lanczos, gd).", type);
        }
        if (no_symm) SETERRQ(mpi_comm, PETSC_ERR_SUP, "Unsupported option: no_symm.");
        /* the explicit (assembled MATMPIAIJ) superblock Hamiltonian of the reference, -do_shell 0, is not built here: the matrix-free
           path is the product; asked for, it is refused rather than silently ignored */
        if (!do_shell) SETERRQ(mpi_comm, PETSC_ERR_SUP, "Unsupported option: -do_shell 0 (only the matrix-free superblock Hamiltonian is implemented).");
        ierr = PetscOptionsGetBool(NULL, NULL, "-debug_check_symmetry", &debug_symm, NULL); CHKERRQ(ierr);
        ierr = PetscOptionsGetBool(NULL, NULL, "-wavefunction_guess", &use_guess, NULL); CHKERRQ(ierr);
        ierr = PetscOptionsGetBool(NULL, NULL, "-wavefunction_guess_overlap", &use_guess_overlap, NULL); CHKERRQ(ierr);
        ierr = PetscOptionsGetBool(NULL, NULL, "-rdm_warm_start", &use_rdm_warm, NULL); CHKERRQ(ierr);
        ierr = PetscOptionsGetBool(NULL, NULL, "-corr_batch", &use_corr_batch, NULL); CHKERRQ(ierr);
        ierr = PetscOptionsGetBool(NULL, NULL, "-prune_ops", &prune_ops, NULL); CHKERRQ(ierr);
        ierr = PetscOptionsGetBool(NULL, NULL, "-step_profile", &step_profile, NULL); CHKERRQ(ierr);

        PetscBool opt = PETSC_FALSE;
        ierr = PetscOptionsGetString(NULL, NULL, "-scratch_dir", path, PETSC_MAX_PATH_LEN, &opt); CHKERRQ(ierr);
        scratch_dir = opt ? std::string(path) : std::string("./scratch_dir/");
        if (scratch_dir.back() != '/') scratch_dir += '/';
        do_scratch_dir = opt;        /* blocks live in HBM (288 GB): checkpoints are written only when -scratch_dir is given */
        if (do_scratch_dir && !mpi_rank) { ierr = Makedir(scratch_dir); CHKERRQ(ierr); }
        if (restart && num_sites != restart_num_sites) SETERRQ2(mpi_comm, 1, "The restart data is for %lld sites, the model has %lld.", LLD(restart_num_sites), LLD(num_sites));
        ierr = PetscOptionsGetString(NULL, NULL, "-data_dir", path, PETSC_MAX_PATH_LEN, &opt); CHKERRQ(ierr);
        data_dir = opt ? std::string(path) : std::string("./data_dir/");
        if (data_dir.back() != '/') data_dir += '/';
        if (!mpi_rank) { ierr = Makedir(data_dir); CHKERRQ(ierr); }
        ierr = PetscFOpen(mpi_comm, (data_dir + "DMRGSteps.json").c_str(), "w", &fp_step); CHKERRQ(ierr);
        ierr = SaveStepHeaders(); CHKERRQ(ierr);
        fprintf(fp_step, "[\n");
        ierr = PetscFOpen(mpi_comm, (data_dir + "Timings.json").c_str(), "w", &fp_timings); CHKERRQ(ierr);
        ierr = SaveTimingsHeaders(); CHKERRQ(ierr);
        fprintf(fp_timings, "[\n");
        ierr = PetscFOpen(mpi_comm, (data_dir + "EntanglementSpectra.json").c_str(), "w", &fp_entanglement); CHKERRQ(ierr);
        fprintf(fp_entanglement, "[\n");
        ierr = PetscFOpen(mpi_comm, (data_dir + "Correlations.json").c_str(), "w", &fp_corr); CHKERRQ(ierr);
        ierr = PetscFOpen(mpi_comm, (data_dir + "DMRGRun.json").c_str(), "w", &fp_data); CHKERRQ(ierr);
        ierr = PetscFOpen(mpi_comm, (data_dir + "KronStats.json").c_str(), "w", &fp_kron); CHKERRQ(ierr);
        fprintf(fp_kron, "[\n");
        fprintf(fp_data, "{\n"); Ham.SaveOut(fp_data); fprintf(fp_data, ",\n  \"QNSector\": %g", qn_sector); fflush(fp_data);

        if (!mpi_rank) {
            printf("=========================================\nDENSITY MATRIX RENORMALIZATION GROUP (MI355X engine)\n-----------------------------------------\n");
            Ham.PrintOut();
            printf("-----------------------------------------\nDIRECTORIES\n  Data:    %s\n=========================================\n", data_dir.c_str());
        }
        {   /* sweep modes: -nsweeps | -msweeps [-maxnsweeps] */
            PetscBool opt_mstates = PETSC_FALSE, opt_mwarmup = PETSC_FALSE, opt_nsweeps = PETSC_FALSE, opt_msweeps = PETSC_FALSE, opt_maxnsweeps = PETSC_FALSE;
            PetscInt mstates = 0, num_msweeps = 1000, num_maxnsweeps = 1000;
            msweeps.resize(num_msweeps); maxnsweeps.resize(num_maxnsweeps);
            ierr = PetscOptionsGetInt(NULL, NULL, "-mstates", &mstates, &opt_mstates); CHKERRQ(ierr);
            ierr = PetscOptionsGetInt(NULL, NULL, "-mwarmup", &mwarmup, &opt_mwarmup); CHKERRQ(ierr);
            ierr = PetscOptionsGetInt(NULL, NULL, "-nsweeps", &nsweeps, &opt_nsweeps); CHKERRQ(ierr);
            ierr = PetscOptionsGetIntArray(NULL, NULL, "-msweeps", msweeps.data(), &num_msweeps, &opt_msweeps); CHKERRQ(ierr);
            msweeps.resize(num_msweeps);
            ierr = PetscOptionsGetIntArray(NULL, NULL, "-maxnsweeps", maxnsweeps.data(), &num_maxnsweeps, &opt_maxnsweeps); CHKERRQ(ierr);
            maxnsweeps.resize(num_maxnsweeps);
            if (opt_mstates && !opt_mwarmup) mwarmup = mstates;
            if (opt_nsweeps && opt_msweeps) SETERRQ(mpi_comm, 1, "-msweeps and -nsweeps cannot both be specified at the same time.");
            if (opt_maxnsweeps && (num_maxnsweeps != num_msweeps))
                SETERRQ2(mpi_comm, 1, "-msweeps and -maxnsweeps must have the same number of items. Got %lld and %lld, respectively.", LLD(num_msweeps), LLD(num_maxnsweeps));
            if (opt_nsweeps && !opt_msweeps) sweep_mode = SWEEP_MODE_NSWEEPS;
            else if (opt_msweeps && !opt_nsweeps) sweep_mode = opt_maxnsweeps ? SWEEP_MODE_TOLERANCE_TEST : SWEEP_MODE_MSWEEPS;
            else sweep_mode = SWEEP_MODE_NULL;
            /* engine extension: the MinBlock argument of SingleSweep (reference: include/DMRGBlockContainer.hpp:996-1013, always left
               at its default of 1 by the reference's driver) from the command line -- sweeps that turn round min_block sites before
               the edge never truncate against a rank-deficient density matrix of a few-site environment, which is what lets the
               parity tests compare runs at m = 24-48 with the oracle step by step */
            ierr = PetscOptionsGetInt(NULL, NULL, "-min_block", &opt_min_block, NULL); CHKERRQ(ierr);
        }
        init = PETSC_TRUE;
        return 0;
    }

    /** Registers an n-point correlator, measured on the superblock ground state at the centre of the lattice at the end
        of the warm-up and of every sweep.  Sites are numbered on the superblock; an operator on the right half is
        carried over to the environment block by reflection (include/DMRGBlockContainer.hpp:627-682 of the reference). */
    PetscErrorCode SetUpCorrelation(const std::vector<Op>& OpList, const std::string& name, const std::string& desc)
    {
        if (!init) SETERRQ(mpi_comm, 1, "Initialize() must be called first.");
        if (LoopType == SweepStep) SETERRQ(mpi_comm, 1, "Setup correlation functions should be called before starting the sweeps.");
        Correlator m;
        m.idx = (PetscInt)measurements.size();
        m.name = name; m.desc1 = desc;
        m.desc2 += "< ";
        for (const Op& op : OpList) m.desc2 += OpToStr(op.OpType) + "_{" + std::to_string(op.idx) + "} ";
        m.desc2 += ">";
        for (const Op& op : OpList) {
            if (0 <= op.idx && op.idx < num_sites / 2) m.SysOps.push_back(op);
            else if (num_sites / 2 <= op.idx && op.idx < num_sites) m.EnvOps.push_back({op.OpType, num_sites - 1 - op.idx});
            else SETERRQ2(mpi_comm, 1, "Operator index must be in the range [0,%lld). Got %lld.", LLD(num_sites), LLD(op.idx));
        }
        if (m.SysOps.empty()) { m.SysOps = m.EnvOps; m.EnvOps.clear(); }      /* reflection symmetry */
        m.desc3 += "< ( ";
        for (const Op& op : m.SysOps) m.desc3 += OpToStr(op.OpType) + "_{" + std::to_string(op.idx) + "} ";
        if (m.SysOps.empty()) m.desc3 += "1 ";
        m.desc3 += ") (x) ( ";
        for (const Op& op : m.EnvOps) m.desc3 += OpToStr(op.OpType) + "_{" + std::to_string(op.idx) + "} ";
        if (m.EnvOps.empty()) m.desc3 += "1 ";
        m.desc3 += ") >";
        measurements.push_back(m);
        need_built = false;             /* the residency tables depend on the registered correlators */
        return 0;
    }

    /** Grows the system block from one site to half the lattice, using as environment the largest stored block that
        completes whole columns (Liang-Pang style cluster growth). */
    PetscErrorCode Warmup()
    {
        if (!init) SETERRQ(mpi_comm, 1, "Initialize() must be called first.");
        if (dry_run) return 0;
        if (mwarmup == 0) { if (!mpi_rank) std::cout << "WARNING: Nothing left to do since mwarmup is zero." << std::endl; return 0; }
        PetscErrorCode ierr;
        ierr = PetscTime(&t0abs); CHKERRQ(ierr);
        if (warmed_up) SETERRQ(mpi_comm, 1, "Warmup has already been called, and it can only be called once.");
        if (restart) {   /* the warm-up is replaced by loading the checkpointed blocks (reference :737-757) */
            if (!mpi_rank) printf("Loading blocks from file...\n");
            num_sys_blocks = num_sites - 1;
            sys_blocks.resize((size_t)num_sys_blocks);
            for (PetscInt ib = 0; ib < restart_sys_ninit; ++ib) {
                ierr = sys_blocks[(size_t)ib].InitializeFromDisk(mpi_comm, restart_dir + BlockDir("Sys", ib)); CHKERRQ(ierr);
            }
            sys_ninit = restart_sys_ninit;
            warmed_up = PETSC_TRUE;
            ierr = PetscTime(&t0abs); CHKERRQ(ierr);
            return 0;
        }
        if (!mpi_rank) printf("WARMUP\n");
        num_sys_blocks = num_sites - 1;
        sys_blocks.resize((size_t)num_sys_blocks);
        ierr = sys_blocks[sys_ninit++].Initialize(mpi_comm, 1, PETSC_DEFAULT); CHKERRQ(ierr);
        if (AddSite().NumSites() != 1) SETERRQ1(mpi_comm, 1, "Routine assumes an additional site of 1. Got %lld.", LLD(AddSite().NumSites()));
        PetscInt nsites_cluster = Ham.NumEnvSites();
        if (nsites_cluster % 2) nsites_cluster *= 2;
        if (!mpi_rank) printf(" Preparing initial blocks.\n");
        while (sys_ninit < nsites_cluster) {          /* exact blocks, no truncation */
            const PetscInt ntot = sys_blocks[sys_ninit - 1].NumSites() + AddSite().NumSites();
            ierr = KronEye_Explicit(sys_blocks[sys_ninit - 1], AddSite(), Ham.H(ntot), sys_blocks[sys_ninit]); CHKERRQ(ierr);
            ++sys_ninit;
        }
        if (sys_ninit >= num_sites / 2)
            SETERRQ(mpi_comm, 1, "No DMRG Steps were performed since all site operators were created exactly.  Please change the system dimensions.");
        LoopType = WarmupStep; StepIdx = 0;
        while (sys_ninit < num_sites / 2) {
            PetscInt full_cluster = (((sys_ninit + 2) / nsites_cluster) + 1) * nsites_cluster;
            PetscInt env_numsites = full_cluster - sys_ninit - 2;
            const PetscInt env_add = ((sys_ninit - env_numsites) / nsites_cluster) * nsites_cluster;
            env_numsites += env_add; full_cluster += env_add;
            if (env_numsites < 1 || env_numsites > sys_ninit) SETERRQ1(mpi_comm, 1, "Incorrect number of sites. Got %lld.", LLD(env_numsites));
            if (!mpi_rank) { printf(" %s  %lld/%lld/%lld\n", "WARMUP", LLD(LoopIdx), LLD(StepIdx), LLD(GlobIdx)); PrintBlocks(sys_ninit, env_numsites); }
            /* operator residency (engine extension, -prune_ops 0 disables): the new system block carries the sites a later
               inter-block term or a registered correlator can touch; the re-derived environment block only coupling sites */
            StepHints hints;
            if (prune_ops) {
                hints.keep_sys = NeedMask(sys_ninit + 1, true, false);
                hints.keep_env = NeedMask(env_numsites + 1, env_numsites == sys_ninit, false);
            }
            ierr = SingleDMRGStep(sys_blocks[sys_ninit - 1], sys_blocks[env_numsites - 1], mwarmup,
                                  sys_blocks[sys_ninit], sys_blocks[env_numsites], PetscBool(sys_ninit + 1 == num_sites / 2), &hints); CHKERRQ(ierr);
            if (prune_ops) { ierr = PruneConsumed(sys_ninit - 1); CHKERRQ(ierr); }
            ++sys_ninit;
        }
        if (sys_ninit != num_sites / 2) SETERRQ2(mpi_comm, 1, "Expected sys_ninit = num_sites/2 = %lld. Got %lld.", LLD(num_sites / 2), LLD(sys_ninit));
        warmed_up = PETSC_TRUE;
        ierr = SaveSweepsData(); CHKERRQ(ierr);
        ++LoopIdx;
        return 0;
    }

    PetscErrorCode Sweeps()
    {
        if (dry_run || mwarmup == 0) return 0;
        PetscErrorCode ierr;
        const PetscInt first = (restart && restart_options) ? restart_msweep_idx + 1 : 0;      /* continue the saved schedule */
        if (sweep_mode == SWEEP_MODE_NSWEEPS) { for (msweep_idx = first; msweep_idx < nsweeps; ++msweep_idx) { ierr = SingleSweep(mwarmup, opt_min_block); CHKERRQ(ierr); } }
        else if (sweep_mode == SWEEP_MODE_MSWEEPS) { for (msweep_idx = first; msweep_idx < (PetscInt)msweeps.size(); ++msweep_idx) { ierr = SingleSweep(msweeps.at(msweep_idx), opt_min_block); CHKERRQ(ierr); } }
        else if (sweep_mode == SWEEP_MODE_TOLERANCE_TEST) {
            for (msweep_idx = first; msweep_idx < (PetscInt)msweeps.size(); ++msweep_idx) {
                const PetscInt mstates = msweeps.at(msweep_idx), max_iter = maxnsweeps.at(msweep_idx);
                if (max_iter == 0) continue;
                PetscInt iter = 0; bool cont;
                do {    /* sweep again while the energy still moves by more than the largest truncation error */
                    const PetscScalar prev_gse = gse;
                    ierr = SingleSweep(mstates, opt_min_block); CHKERRQ(ierr);
                    const PetscReal diff_gse = PetscAbsScalar(gse - prev_gse);
                    const PetscReal max_trn = std::max(*std::max_element(trunc_err.begin(), trunc_err.end()), 0.0);
                    ++iter;
                    cont = (iter < max_iter) && (diff_gse > max_trn);
                    if (!mpi_rank) std::cout << "SWEEP_MODE_TOLERANCE_TEST\n  Iterations / Max Iterations:       " << iter << "/" << max_iter
                        << "\n  Difference in ground state energy: " << diff_gse << "\n  Largest truncation error:          " << max_trn
                        << "\n  " << (cont ? "CONTINUE" : "BREAK") << std::endl;
                } while (cont);
            }
        }
        else if (sweep_mode == SWEEP_MODE_NULL) {}
        else SETERRQ(mpi_comm, 1, "Invalid sweep mode.");
        return 0;
    }

    /** One sweep: centre -> right edge, then (by reflection symmetry) from the right edge's mirror back to the centre:
        N-4 steps. */
    PetscErrorCode SingleSweep(const PetscInt& MStates, const PetscInt& MinBlock = PETSC_DEFAULT)
    {
        if (!init) SETERRQ(mpi_comm, 1, "Initialize() must be called first.");
        PetscErrorCode ierr;
        if (!warmed_up) SETERRQ(mpi_comm, 1, "Warmup must be called first before performing sweeps.");
        if (!mpi_rank) printf("SWEEP MStates=%lld\n", LLD(MStates));
        trunc_err.clear();
        const PetscInt min_block = MinBlock == PETSC_DEFAULT ? 1 : MinBlock;
        if (min_block < 1) SETERRQ1(mpi_comm, 1, "MinBlock must at least be 1. Got %lld.", LLD(min_block));
        PetscLogDouble ts0, ts1;
        ierr = PetscTime(&ts0); CHKERRQ(ierr);
        const PetscInt steps0 = GlobIdx, mm0 = total_matmults, rot0 = rot_ops_total;
        LoopType = SweepStep; StepIdx = 0;
        for (PetscInt iblock = num_sites / 2; iblock < num_sites - min_block - 2; ++iblock) {
            const PetscInt insys = iblock - 1, inenv = num_sites - iblock - 3, outsys = iblock, outenv = num_sites - iblock - 2;
            if (!mpi_rank) { printf(" %s  %lld/%lld/%lld\n", "SWEEP", LLD(LoopIdx), LLD(StepIdx), LLD(GlobIdx)); PrintBlocks(insys + 1, inenv + 1); }
            /* centre -> edge: the growing block is read again as a LEFT block only; the re-derived shrinking block
               (index outenv) is overwritten by the way back before any step reads it: its operators are not computed */
            StepHints hints;
            if (prune_ops) { hints.keep_sys = NeedMask(outsys + 1, false, true); hints.dead_env = true; }
            ierr = SingleDMRGStep(sys_blocks[insys], sys_blocks[inenv], MStates, sys_blocks[outsys], sys_blocks[outenv], PETSC_FALSE, &hints); CHKERRQ(ierr);
        }
        for (PetscInt iblock = min_block; iblock < num_sites / 2; ++iblock) {
            const PetscInt insys = num_sites - iblock - 3, inenv = iblock - 1, outsys = num_sites - iblock - 2, outenv = iblock;
            if (!mpi_rank) { printf(" %s  %lld/%lld/%lld\n", "SWEEP", LLD(LoopIdx), LLD(StepIdx), LLD(GlobIdx)); PrintBlocks(insys + 1, inenv + 1); }
            /* edge -> centre: the small block grows towards the measurement at the centre (coupling + correlator sites); the
               re-derived large block (index outsys) is overwritten by the next sweep before any step reads it */
            StepHints hints;
            if (prune_ops) {
                if (outsys == outenv) hints.keep_sys = NeedMask(outsys + 1, false, false);
                else { hints.keep_env = NeedMask(outenv + 1, true, false); hints.dead_sys = true; }
            }
            ierr = SingleDMRGStep(sys_blocks[insys], sys_blocks[inenv], MStates, sys_blocks[outsys], sys_blocks[outenv], PetscBool(outsys == outenv), &hints); CHKERRQ(ierr);
            if (prune_ops) { ierr = PruneConsumed(inenv); CHKERRQ(ierr); }
        }
        sweeps_mstates.push_back(MStates);
        ierr = PetscTime(&ts1); CHKERRQ(ierr);
        if (dmrgx_host::WorldComm() && dmrgx_host::WorldSize() > 1) {      /* every rank: its own share of the rotation work (the correlators are dealt over the ranks) */
            printf("[rank %d] SWEEP rotated operators = %lld\n", dmrgx_host::WorldRank(), LLD(rot_ops_total - rot0)); fflush(stdout);
        }
        if (!mpi_rank) printf("SWEEP DONE  steps=%lld  time=%.6f s  sites/s=%.3f  MatMults=%lld  E=%.12g\n", LLD(GlobIdx - steps0), ts1 - ts0,
                              (GlobIdx - steps0) / (ts1 - ts0), LLD(total_matmults - mm0), gse);
        last_sweep_seconds = ts1 - ts0; last_sweep_steps = GlobIdx - steps0; last_sweep_matmults = total_matmults - mm0;
        dmrgx_mem_stats(&device_bytes_after_sweep, nullptr, nullptr);      /* what stays resident between sweeps: the stored blocks */
        ierr = SaveSweepsData(); CHKERRQ(ierr);
        ++LoopIdx;
        return 0;
    }

    PetscErrorCode Destroy()
    {
        if (!init) return 0;
        if (rdm_pending) { dmrgx_rdm* r = rdm_pending; rdm_pending = nullptr; if (dmrgx_rdm_destroy(r)) SETERRQ1(mpi_comm, 1, "dmrgx_rdm_destroy: %s", dmrgx_last_error()); }
        for (Block& b : sys_blocks) { PetscErrorCode ierr = b.Destroy(); CHKERRQ(ierr); }
        PetscErrorCode ierr = SingleSite.Destroy(); CHKERRQ(ierr);
        if (fp_step) { fprintf(fp_step, "\n  ]\n}\n"); fclose(fp_step); fp_step = NULL; }
        if (fp_timings) { fprintf(fp_timings, "\n  ]\n}\n"); fclose(fp_timings); fp_timings = NULL; }
        spectra_writer.Drain();
        if (fp_entanglement) { fprintf(fp_entanglement, "\n]\n"); fclose(fp_entanglement); fp_entanglement = NULL; }
        if (fp_kron) { fprintf(fp_kron, "\n]\n"); fclose(fp_kron); fp_kron = NULL; }
        if (fp_corr) {
            if (!corr_headers_printed) { PetscErrorCode e2 = PrintCorrelationHeaders(); CHKERRQ(e2); }
            fprintf(fp_corr, "\n  ]\n}\n"); fclose(fp_corr); fp_corr = NULL;
        }
        if (fp_data) {
            fprintf(fp_data, ",\n  \"Sweeps\": [");
            for (size_t i = 0; i < sweeps_mstates.size(); ++i) fprintf(fp_data, "%s%lld", i ? ", " : "", LLD(sweeps_mstates[i]));
            size_t in_use = 0, cached = 0, peak = 0;
            dmrgx_mem_stats(&in_use, &cached, &peak);
            fprintf(fp_data, "],\n  \"GSEnergy\": %.16g,\n  \"MatMults\": %lld,\n  \"LastSweepSeconds\": %.9g,\n  \"LastSweepSteps\": %lld,\n  \"LastSweepMatMults\": %lld,\n  \"EigensolveSeconds\": %.9g,\n"
                             "  \"DeviceBytesResidentAfterSweep\": %zu,\n  \"DeviceBytesPeak\": %zu,\n  \"DeviceBytesCached\": %zu,\n  \"StartVectorsTransformed\": %lld,\n  \"StartVectorsThroughOverlap\": %lld,\n  \"StartVectorsRejected\": %lld,\n"
                             "  \"RdmCalls\": %lld,\n  \"RdmBlockJacobiCalls\": %lld,\n  \"TridPersistentCalls\": %lld,\n  \"TridLaunchPathCalls\": %lld,\n  \"TridFallbacks\": %lld,\n"
                             "  \"TridMaxWorkgroupsPerMatrix\": %lld,\n  \"RdmMaxMergeLevels\": %lld,\n  \"RdmMaxWyBlocks\": %lld,\n  \"Ranks\": %d\n}\n",
                    gse, LLD(total_matmults), last_sweep_seconds, LLD(last_sweep_steps), LLD(last_sweep_matmults), total_eigs_seconds, device_bytes_after_sweep, peak, cached, LLD(guesses_used), LLD(guesses_projected), LLD(guesses_rejected),
                    LLD(rdm_calls), LLD(rdm_jacobi_calls), LLD(trid_persistent_calls), LLD(trid_launch_calls), LLD(trid_fallbacks), LLD(trid_max_wgs), LLD(rdm_max_levels), LLD(rdm_max_wy_blocks), (int)mpi_size);
            fclose(fp_data); fp_data = NULL;
        }
        init = PETSC_FALSE;
        return 0;
    }

    const Block& SysBlock(const PetscInt& BlockIdx) const { if (BlockIdx >= sys_ninit) throw std::runtime_error("Attempted to access uninitialized system block."); return sys_blocks[BlockIdx]; }
    const Block& EnvBlock() const { return sys_blocks[0]; }
    PetscInt NumSites() const { return num_sites; }
    const Hamiltonian& HamiltonianRef() const { return Ham; }
    PetscBool Verbose() const { return verbose; }
    PetscScalar GSEnergy() const { return gse; }
    PetscInt TotalMatMults() const { return total_matmults; }

    /** One DMRG step: enlarge both blocks by a site, solve the superblock ground state in the target sector, truncate
        to at most MStates states per block and rotate every operator into the new bases. */
    /** Residency hints of the sweep schedule for the two output blocks of a step (engine extension; the reference keeps
        every Sz(i)/Sp(i) of every block and spills whole blocks to disk, src/DMRGBlock.cpp:1090-1103): which site operators
        to rotate (empty = all), or that the block is never read again (only its sector table is recorded). */
    struct StepHints { std::vector<char> keep_sys, keep_env; bool dead_sys = false, dead_env = false; };

    PetscErrorCode SingleDMRGStep(Block& SysBlock, Block& EnvBlock, const PetscInt& MStates, Block& SysBlockOut, Block& EnvBlockOut,
                                  PetscBool do_measurements = PETSC_FALSE, const StepHints* hints = nullptr)
    {
        PetscErrorCode ierr;
        PetscLogDouble t0 = t0abs, tenlr, tkron, tdiag, trdms, trotb;
        TimingsData timings;
        StepData step;
        step.NumSites_Sys = SysBlock.NumSites(); step.NumSites_Env = EnvBlock.NumSites();
        step.NumStates_Sys = SysBlock.NumStates(); step.NumStates_Env = EnvBlock.NumStates();
        const PetscBool same = PetscBool(&SysBlock == &EnvBlock);

        Block SysBlockEnl, EnvBlockEnl;
        ierr = KronEye_Explicit(SysBlock, AddSite(), Ham.H(SysBlock.NumSites() + AddSite().NumSites()), SysBlockEnl); CHKERRQ(ierr);
        if (!same) { ierr = KronEye_Explicit(EnvBlock, AddSite(), Ham.H(EnvBlock.NumSites() + AddSite().NumSites()), EnvBlockEnl); CHKERRQ(ierr); }
        else EnvBlockEnl = SysBlockEnl;                                   /* shallow copy: same operator handles */
        ierr = PetscTime(&tenlr); CHKERRQ(ierr);
        timings.tEnlr = tenlr - t0;
        step.NumSites_SysEnl = SysBlockEnl.NumSites(); step.NumSites_EnvEnl = EnvBlockEnl.NumSites();
        step.NumStates_SysEnl = SysBlockEnl.NumStates(); step.NumStates_EnvEnl = EnvBlockEnl.NumStates();

        const PetscInt NumSitesTotal = SysBlockEnl.NumSites() + EnvBlockEnl.NumSites();
        const std::vector<Hamiltonians::Term> Terms = Ham.H(NumSitesTotal);
        KronBlocks_t KronBlocks(SysBlockEnl, EnvBlockEnl, {qn_sector}, NULL, GlobIdx);
        step.NumStates_H = KronBlocks.NumStates();
        if (KronBlocks.NumStates() == 0) SETERRQ1(mpi_comm, 1, "The target sector %g is empty.", qn_sector);
        Mat H = nullptr;
        ierr = KronBlocks.KronSumSetRedistribute(PETSC_TRUE); CHKERRQ(ierr);
        ierr = KronBlocks.KronSumSetToleranceFromOptions(); CHKERRQ(ierr);
        ierr = KronBlocks.KronSumSetShellMatrix(do_shell); CHKERRQ(ierr);
        ierr = KronBlocks.KronSumConstruct(Terms, H); CHKERRQ(ierr);
        if (!H) SETERRQ(mpi_comm, 1, "H is null.");
        ierr = PetscTime(&tkron); CHKERRQ(ierr);
        timings.tKron = tkron - tenlr;
        dmrgx_kron_info kinfo;
        memset(&kinfo, 0, sizeof(kinfo));
        if (dmrgx_kron_plan_info(H->plan, &kinfo)) SETERRQ1(mpi_comm, 1, "dmrgx_kron_plan_info: %s", dmrgx_last_error());
        if (step_profile && dmrgx_kron_plan_timing(H->plan, 1)) SETERRQ1(mpi_comm, 1, "dmrgx_kron_plan_timing: %s", dmrgx_last_error());

        if (debug_symm) {   /* -debug_check_symmetry: <u,Hv> == <Hu,v> on random vectors (the reference's H is symmetric) */
            Vec u, v, Hu, Hv;
            MatCreateVecs(H, &u, &v); MatCreateVecs(H, &Hu, &Hv);
            double* uh = u->buf->host(); double* vh = v->buf->host();
            uint64_t z = 88172645463325252ull;
            for (PetscInt i = 0; i < u->n; ++i) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; uh[i] = (double)(z % 2001) / 1000.0 - 1.0; z ^= z << 13; z ^= z >> 7; z ^= z << 17; vh[i] = (double)(z % 2001) / 1000.0 - 1.0; }
            ierr = MatMult(H, u, Hu); CHKERRQ(ierr); ierr = MatMult(H, v, Hv); CHKERRQ(ierr);
            const double *a = u->buf->host_ro(), *b = v->buf->host_ro(), *ha = Hu->buf->host_ro(), *hb = Hv->buf->host_ro();
            double uHv = 0, Huv = 0, nu = 0;
            for (PetscInt i = 0; i < u->n; ++i) { uHv += a[i] * hb[i]; Huv += ha[i] * b[i]; nu += ha[i] * ha[i]; }
            printf("  [debug] symmetry: <u,Hv>=%.12g <Hu,v>=%.12g |Hu|=%.6g  N=%lld\n", uHv, Huv, std::sqrt(nu), LLD(u->n));
        }
        /* ground state: lowest eigenpair, random start vector (the reference passes no initial space either) */
        Vec gsv_r;
        ierr = MatCreateVecs(H, &gsv_r, nullptr); CHKERRQ(ierr);
        PetscScalar gse_r = 0.0;
        {
            dmrgx_eigs_opts o;
            memset(&o, 0, sizeof(o));
            o.ncv = (int32_t)eps_ncv; o.max_it = (int32_t)eps_max_it; o.tol = eps_tol; o.seed = 0x9E3779B9u + (uint64_t)GlobIdx;
            o.method = eps_method; o.gd_minv = (int32_t)eps_gd_minv;
            bool guessed = false;
            double min_norm2 = 0.0;
            try { ierr = TransformedGuess(KronBlocks, SysBlock, EnvBlock, gsv_r, guessed, min_norm2); CHKERRQ(ierr); }
            catch (const std::exception& e) { SETERRQ1(mpi_comm, 1, "wavefunction transformation: %s", e.what()); }
            if (guessed) { o.use_initial = 1; o.min_initial_norm2 = min_norm2; }
            dmrgx_eigs_stats st;
            memset(&st, 0, sizeof(st));
            /* -step_profile 1: the solver's own wall time (KronStats.json: eigs_seconds) must not contain the plan's operator copies and the
               start-vector GEMMs still queued in front of it -- bench.py's non-GEMM cost per MatMult is eigs_seconds minus the GEMM events */
            if (step_profile) dmrgx_stream_sync(nullptr);
            if (H->plan_world > 1) {
                /* striped solve: every rank owns a stripe of the right index of every KronBlock; the solver issues one
                   RCCL all-gather per MatMult and two fused all-reduces per Lanczos step itself.  The start vector goes in, and
                   the ground state comes back, in the reference's vector layout, replicated on every rank. */
                dmrgx_host::DevBuffer psi_full((size_t)kinfo.vec_len, dmrgx_host::DevBuffer::device_only_t{});
                if (dmrgx_memset_zero(psi_full.dev_uninitialised(), (size_t)kinfo.vec_len * sizeof(double), nullptr)) SETERRQ1(mpi_comm, 1, "%s", dmrgx_last_error());
                if (guessed && dmrgx_kron_vec_to_striped(H->plan, gsv_r->buf->dev_ro(), psi_full.dev_uninitialised(), nullptr)) SETERRQ1(mpi_comm, 1, "%s", dmrgx_last_error());
                o.comm = dmrgx_host::WorldComm();
                if (dmrgx_eigs_lowest(H->plan, &o, &gse_r, psi_full.dev_uninitialised(), &st, nullptr)) SETERRQ1(mpi_comm, 1, "dmrgx_eigs_lowest: %s", dmrgx_last_error());
                if (dmrgx_kron_vec_from_striped(H->plan, psi_full.dev_ro(), gsv_r->buf->dev_uninitialised(), nullptr)) SETERRQ1(mpi_comm, 1, "%s", dmrgx_last_error());
            }
            else if (dmrgx_eigs_lowest(H->plan, &o, &gse_r, gsv_r->buf->dev_uninitialised(), &st, nullptr))
                SETERRQ1(mpi_comm, 1, "dmrgx_eigs_lowest: %s", dmrgx_last_error());
            if (guessed && st.start_rejected) { ++guesses_rejected; guessed = false; }
            if (guessed) ++guesses_used;
            timings.nMatMult = st.n_matvec; total_matmults += st.n_matvec; total_eigs_seconds += st.seconds;
            double ms4[4] = {0, 0, 0, 0}; int64_t napp = 0;
            if (step_profile && dmrgx_kron_plan_timing_read(H->plan, ms4, &napp)) SETERRQ1(mpi_comm, 1, "dmrgx_kron_plan_timing_read: %s", dmrgx_last_error());
            ierr = SaveKronStats(kinfo, SysBlock, EnvBlock, (PetscInt)Terms.size(), st.n_matvec, st.seconds, ms4, napp); CHKERRQ(ierr);
        }
        step.GSEnergy = gse_r;
        ierr = MatDestroy_KronSumShell(&H); CHKERRQ(ierr);
        ierr = MatDestroy(&H); CHKERRQ(ierr);
        ierr = PetscTime(&tdiag); CHKERRQ(ierr);
        timings.tDiag = tdiag - tkron;

        BasisTransformation BT_L, BT_R;
        /* a side whose output block no later step reads needs its spectrum (truncation error, sector table) but no eigenvectors */
        const bool vec_sys = !(hints && hints->dead_sys && !same), vec_env = !(hints && hints->dead_env && !same);
        ierr = GetTruncation(KronBlocks, gsv_r, MStates, BT_L, BT_R, BlockIndex(SysBlockOut), BlockIndex(EnvBlockOut), vec_sys, vec_env); CHKERRQ(ierr);
        ierr = CalculateCorrelations_BlockDiag(KronBlocks, gsv_r, do_measurements); CHKERRQ(ierr);
        if (use_guess) {   /* what the next step needs to carry this ground state over */
            prev.valid = false;
            prev.insys = BlockIndex(SysBlock); prev.inenv = BlockIndex(EnvBlock); prev.outsys = BlockIndex(SysBlockOut); prev.outenv = BlockIndex(EnvBlockOut);
            if (prev.insys >= 0 && prev.inenv >= 0 && prev.outsys >= 0 && prev.outenv >= 0 && BT_L.RotMatT && BT_R.RotMatT) {
                prev.psi = gsv_r;
                prev.kb.clear();
                for (PetscInt k = 0; k < KronBlocks.size(); ++k)
                    prev.kb.push_back({KronBlocks.LeftIdx(k), KronBlocks.RightIdx(k), SysBlockEnl.Magnetization.Sizes(KronBlocks.LeftIdx(k)),
                                       EnvBlockEnl.Magnetization.Sizes(KronBlocks.RightIdx(k)), KronBlocks.Offsets(k)});
                ierr = EnlargedParts(SysBlock, prev.partsL); CHKERRQ(ierr);
                ierr = EnlargedParts(EnvBlock, prev.partsR); CHKERRQ(ierr);
                prev.rotL = BT_L.RotMatT->rot; prev.rotR = BT_R.RotMatT->rot;
                if (block_rot.size() < sys_blocks.size()) { block_rot.resize(sys_blocks.size()); rot_meta.resize(sys_blocks.size()); block_ver.resize(sys_blocks.size(), 0); block_ovl.resize(sys_blocks.size()); }
                const int64_t pv_sys = block_ver[(size_t)prev.insys], pv_env = block_ver[(size_t)prev.inenv];      /* (read before either output index is re-versioned) */
                try {
                    ierr = RecordBlockBasis(prev.outsys, prev.rotL, pv_sys, prev.partsL, SysBlock); CHKERRQ(ierr);
                    if (!same) { ierr = RecordBlockBasis(prev.outenv, prev.rotR, pv_env, prev.partsR, EnvBlock); CHKERRQ(ierr); }
                } catch (const std::exception& e) { SETERRQ1(mpi_comm, 1, "basis overlap: %s", e.what()); }
                prev.valid = true;
            }
        }
        ierr = VecDestroy(&gsv_r); CHKERRQ(ierr);
        ierr = SysBlockOut.Destroy(); CHKERRQ(ierr);
        ierr = EnvBlockOut.Destroy(); CHKERRQ(ierr);
        ierr = PetscTime(&trdms); CHKERRQ(ierr);
        timings.tRdms = trdms - tdiag;

        ierr = SysBlockOut.Initialize(SysBlockEnl.NumSites(), BT_L.QN); CHKERRQ(ierr);
        if (hints && hints->dead_sys && !same) SysBlockOut.MarkDead();
        else { ierr = SysBlockOut.RotateOperators(SysBlockEnl, BT_L.RotMatT, hints && !hints->keep_sys.empty() ? &hints->keep_sys : nullptr); CHKERRQ(ierr); }
        if (!same) {
            ierr = EnvBlockOut.Initialize(EnvBlockEnl.NumSites(), BT_R.QN); CHKERRQ(ierr);
            if (hints && hints->dead_env) EnvBlockOut.MarkDead();
            else { ierr = EnvBlockOut.RotateOperators(EnvBlockEnl, BT_R.RotMatT, hints && !hints->keep_env.empty() ? &hints->keep_env : nullptr); CHKERRQ(ierr); }
        }
        timings.nRotOps = (SysBlockOut.Dead() ? 0 : SysBlockOut.NumRotatedOps()) + ((same || EnvBlockOut.Dead()) ? 0 : EnvBlockOut.NumRotatedOps());
        rot_ops_total += timings.nRotOps;
        step.NumStates_SysRot = SysBlockOut.NumStates(); step.NumStates_EnvRot = EnvBlockOut.NumStates();
        step.TruncErr_Sys = BT_L.TruncErr; step.TruncErr_Env = BT_R.TruncErr;
        ierr = PetscTime(&trotb); CHKERRQ(ierr);
        timings.tRotb = trotb - trdms;
        timings.Total = trotb - t0;
        ierr = PetscTime(&t0abs); CHKERRQ(ierr);
        if (!mpi_rank && verbose) {
            printf("  Superblock: NumStates %lld  NumSites %lld  QNSector %g  Energy %.12g  Energy/site %.12g\n", LLD(KronBlocks.NumStates()), LLD(NumSitesTotal), qn_sector, gse_r, gse_r / PetscReal(NumSitesTotal));
            printf("  Sys out: NumStates %lld TruncErr %g | Env out: NumStates %lld TruncErr %g | MatMults %lld\n", LLD(BT_L.QN.NumStates()), BT_L.TruncErr, LLD(BT_R.QN.NumStates()), BT_R.TruncErr, LLD(timings.nMatMult));
            printf("  Total %.6f s: enlarge %.6f  build H %.6f  solve %.6f  rdm %.6f  rotate %.6f\n", timings.Total, timings.tEnlr, timings.tKron, timings.tDiag, timings.tRdms, timings.tRotb);
        }
        gse = gse_r;
        trunc_err.push_back(BT_L.TruncErr);
        ierr = SaveStepData(step); CHKERRQ(ierr);
        if (mpi_size > 1) { int64_t nmv = 0; dmrgx_eigs_comm_timing(&timings.msAllGather, &timings.msApply, &nmv, 1); }      /* this step's share, totals reset */
        ierr = SaveTimingsData(timings); CHKERRQ(ierr);
        ++GlobIdx; ++StepIdx; ++rows_written;
        return 0;
    }

    /* ---- wavefunction transformation ------------------------------------------------------------------------------
       The reference starts every eigensolve from a random vector (include/DMRGBlockContainer.hpp:1489-1496; its legacy
       solver seeded it with the previous ground state, old/idmrg.cpp:203).  Here the previous step's ground state is
       carried into the new superblock basis (White 1996): with R the truncation just computed on the growing side and R'
       the stored truncation that created the shrinking side's block from the next smaller one,
           psi'[(a' s_b), (b' s_c)] = sum  R[a'; (a s_a)]  psi[(a s_a), (b s_b)]  R'[b; (b' s_c)],
       two grouped MFMA launches over sector blocks.  Converged results do not depend on the start vector; the number of
       MatMults per step does.  -wavefunction_guess 0 restores the random start. */
    PetscInt BlockIndex(const Block& b) const
    {
        if (sys_blocks.empty()) return -1;
        const Block* first = &sys_blocks[0];
        if (&b < first || &b >= first + sys_blocks.size()) return -1;
        return (PetscInt)(&b - first);
    }
    /** Decomposition of the sectors of `old (x) site` into (old sector, site state) parts, in basis order
        (the merged-KronBlock order of KronEye_Explicit). */
    PetscErrorCode EnlargedParts(Block& old, std::vector<std::vector<EnlPart>>& parts)
    {
        KronBlocks_t KB(old, AddSite(), {}, NULL, -1);
        parts.clear();
        PetscReal last = 0; bool have = false;
        for (PetscInt k = 0; k < KB.size(); ++k) {
            if (!have || KB.QN(k) < last) { parts.emplace_back(); last = KB.QN(k); have = true; }
            int32_t off = 0;
            for (const EnlPart& e : parts.back()) off += e.size;
            parts.back().push_back(EnlPart{(int32_t)KB.LeftIdx(k), (int32_t)KB.RightIdx(k), off, (int32_t)KB.Sizes(k)});
        }
        return 0;
    }
    static int Qn2(PetscReal q) { return (int)std::lround(2.0 * q); }
    static bool HasVectors(const dmrgx_host::BasisRotation& R)
    {
        for (size_t a = 0; a < R.kept.size(); ++a) if (R.kept[a] > 0 && !R.rt[a]) return false;
        return true;
    }
    /** Description of (parent (x) site) for a rotation: sector decomposition, quantum numbers, sizes. */
    PetscErrorCode DescribeEnlarged(Block& parent, const std::vector<std::vector<EnlPart>>& parts, RotMeta& m)
    {
        m.parts = parts;
        m.parent_qn2.clear(); m.enl_qn2.clear(); m.enl_sizes.clear();
        for (PetscReal q : parent.Magnetization.ListRef()) m.parent_qn2.push_back(Qn2(q));
        std::vector<int> site_qn2;
        for (PetscReal q : AddSite().Magnetization.ListRef()) site_qn2.push_back(Qn2(q));
        for (const std::vector<EnlPart>& sec : parts) {
            int32_t n = 0;
            for (const EnlPart& e : sec) n += e.size;
            m.enl_sizes.push_back(n);
            m.enl_qn2.push_back(sec.empty() ? INT_MIN : m.parent_qn2.at((size_t)sec[0].old_sector) + site_qn2.at((size_t)sec[0].site_sector));
        }
        return 0;
    }
    /** R maps (parent (x) site) -> block and is written in the enlarged basis `from` describes.  Returns the same rotation written
        in the enlarged basis of ANOTHER version of the parent (`to`), every (parent sector, site state) part carried over through
        `inner` = <parent version of `from` | parent version of `to`>.  Sectors or parts without a counterpart drop out (zero). */
    std::shared_ptr<dmrgx_host::BasisRotation> ProjectRotation(const dmrgx_host::BasisRotation& R, const RotMeta& from, const BasisOverlap& inner, const RotMeta& to)
    {
        auto out = std::make_shared<dmrgx_host::BasisRotation>();
        const size_t na = R.kept.size();
        out->old_sizes = to.enl_sizes; out->kept = R.kept; out->old_sector.assign(na, -1); out->rt.assign(na, nullptr);
        std::vector<int64_t> off(na, -1);
        int64_t tot = 0;
        for (size_t a = 0; a < na; ++a) {
            if (R.kept[a] <= 0 || !R.rt[a] || R.old_sector[a] < 0) continue;
            const int q2 = from.enl_qn2.at((size_t)R.old_sector[a]);
            for (size_t I = 0; I < to.enl_qn2.size(); ++I) if (to.enl_qn2[I] == q2) { out->old_sector[a] = (int32_t)I; break; }
            if (out->old_sector[a] < 0) continue;
            off[a] = tot; tot += (int64_t)R.kept[a] * to.enl_sizes[(size_t)out->old_sector[a]];
        }
        if (tot == 0) return out;
        auto arena = std::make_shared<dmrgx_host::DevBuffer>((size_t)tot, dmrgx_host::DevBuffer::device_only_t{});
        double* base = arena->dev_uninitialised();
        if (dmrgx_memset_zero(base, (size_t)tot * sizeof(double), nullptr)) throw std::runtime_error(dmrgx_last_error());
        std::vector<dmrgx_gemm_task> tasks;
        for (size_t a = 0; a < na; ++a) {
            if (off[a] < 0) continue;
            const int32_t If = R.old_sector[a], It = out->old_sector[a], nF = from.enl_sizes.at((size_t)If), nT = to.enl_sizes[(size_t)It], ka = R.kept[a];
            /* the sector's rows live in the arena; DevBuffer views are not available, so every sector gets its own small handle on it */
            out->rt[a] = std::make_shared<dmrgx_host::DevBuffer>(arena, (size_t)off[a], (size_t)ka * nT);
            for (const EnlPart& pf : from.parts.at((size_t)If)) {
                if (pf.size == 0) continue;
                const int qj = from.parent_qn2.at((size_t)pf.old_sector);
                auto ic = inner.cells.find(qj);
                if (ic == inner.cells.end() || ic->second.rows != pf.size) continue;
                for (const EnlPart& pt : to.parts[(size_t)It]) {
                    if (pt.site_sector != pf.site_sector || to.parent_qn2.at((size_t)pt.old_sector) != qj || pt.size == 0 || ic->second.cols != pt.size) continue;
                    tasks.push_back(dmrgx_gemm_task{ka, pt.size, pf.size, 0, R.rt[a]->dev_ro() + pf.off, nF, ic->second.buf->dev_ro(), pt.size, base + off[a] + pt.off, nT});
                }
            }
        }
        if (!tasks.empty() && dmrgx_dgemm_batch((int32_t)tasks.size(), tasks.data(), nullptr)) throw std::runtime_error(dmrgx_last_error());
        return out;
    }
    /** <old version | new version> of one block from its two creation rotations written in the SAME enlarged basis:
        O[q] = R_old[q] R_new[q]^T. */
    std::shared_ptr<BasisOverlap> OverlapOfRotations(const dmrgx_host::BasisRotation& RX, const dmrgx_host::BasisRotation& RY, const std::vector<int>& enl_qn2)
    {
        auto O = std::make_shared<BasisOverlap>();
        struct Pair { size_t a, b; int64_t toff, ooff; };
        std::vector<Pair> pairs;
        int64_t ttot = 0, otot = 0;
        for (size_t a = 0; a < RX.kept.size(); ++a) {
            if (RX.kept[a] <= 0 || !RX.rt[a] || RX.old_sector[a] < 0) continue;
            for (size_t b = 0; b < RY.kept.size(); ++b) {
                if (RY.kept[b] <= 0 || !RY.rt[b] || RY.old_sector[b] != RX.old_sector[a]) continue;
                pairs.push_back(Pair{a, b, ttot, otot});
                ttot += (int64_t)RY.kept[b] * RY.old_sizes.at((size_t)RY.old_sector[b]);
                otot += (int64_t)RX.kept[a] * RY.kept[b];
            }
        }
        if (pairs.empty()) return O;
        auto tarena = std::make_shared<dmrgx_host::DevBuffer>((size_t)ttot, dmrgx_host::DevBuffer::device_only_t{});
        auto oarena = std::make_shared<dmrgx_host::DevBuffer>((size_t)otot, dmrgx_host::DevBuffer::device_only_t{});
        if (dmrgx_memset_zero(tarena->dev_uninitialised(), (size_t)ttot * sizeof(double), nullptr)) throw std::runtime_error(dmrgx_last_error());
        std::vector<dmrgx_axpy_task> tr;
        std::vector<dmrgx_gemm_task> gm;
        for (const Pair& p : pairs) {
            const int32_t nE = RY.old_sizes[(size_t)RY.old_sector[p.b]], kx = RX.kept[p.a], ky = RY.kept[p.b];
            if (RX.old_sizes.at((size_t)RX.old_sector[p.a]) != nE) throw std::runtime_error("the two rotations are not written in the same enlarged basis");
            dmrgx_axpy_task t;
            t.dst = tarena->dev_uninitialised() + p.toff; t.dst_base = nullptr; t.src = RY.rt[p.b]->dev_ro(); t.ldd = ky; t.lds = nE; t.nr = nE; t.nc = ky; t.transposed = 1; t.alpha = 1.0;
            tr.push_back(t);
            gm.push_back(dmrgx_gemm_task{kx, ky, nE, 0, RX.rt[p.a]->dev_ro(), nE, tarena->dev_ro() + p.toff, ky, oarena->dev_uninitialised() + p.ooff, ky});
            typename BasisOverlap::Cell c;
            c.rows = kx; c.cols = ky; c.buf = std::make_shared<dmrgx_host::DevBuffer>(oarena, (size_t)p.ooff, (size_t)kx * ky);
            O->cells[enl_qn2.at((size_t)RX.old_sector[p.a])] = c;
        }
        if (dmrgx_cells_axpy((int32_t)tr.size(), tr.data(), nullptr)) throw std::runtime_error(dmrgx_last_error());
        if (dmrgx_dgemm_batch((int32_t)gm.size(), gm.data(), nullptr)) throw std::runtime_error(dmrgx_last_error());
        return O;
    }
    /** <a|c> ~ <a|b><b|c> (exact on the part of a's span that version b holds) */
    std::shared_ptr<BasisOverlap> ComposeOverlaps(const BasisOverlap& ab, const BasisOverlap& bc)
    {
        auto O = std::make_shared<BasisOverlap>();
        std::vector<dmrgx_gemm_task> gm;
        int64_t tot = 0;
        for (const auto& kv : ab.cells) { auto it = bc.cells.find(kv.first); if (it != bc.cells.end() && it->second.rows == kv.second.cols) tot += (int64_t)kv.second.rows * it->second.cols; }
        if (tot == 0) return O;
        auto arena = std::make_shared<dmrgx_host::DevBuffer>((size_t)tot, dmrgx_host::DevBuffer::device_only_t{});
        int64_t off = 0;
        for (const auto& kv : ab.cells) {
            auto it = bc.cells.find(kv.first);
            if (it == bc.cells.end() || it->second.rows != kv.second.cols) continue;
            typename BasisOverlap::Cell c;
            c.rows = kv.second.rows; c.cols = it->second.cols; c.buf = std::make_shared<dmrgx_host::DevBuffer>(arena, (size_t)off, (size_t)c.rows * c.cols);
            gm.push_back(dmrgx_gemm_task{c.rows, c.cols, kv.second.cols, 0, kv.second.buf->dev_ro(), kv.second.cols, it->second.buf->dev_ro(), it->second.cols, arena->dev_uninitialised() + off, c.cols});
            O->cells[kv.first] = c;
            off += (int64_t)c.rows * c.cols;
        }
        if (dmrgx_dgemm_batch((int32_t)gm.size(), gm.data(), nullptr)) throw std::runtime_error(dmrgx_last_error());
        return O;
    }
    /** sys_blocks[k] has just been (re)created by `Rnew` from version `parent_ver` of sys_blocks[k-1] (`parent`, decomposition
        `parts`).  If a stored block k+1 is the child of an earlier version of block k, the overlap <that version | new version> is
        formed -- R_old (O_{k-1} (x) 1) R_new^T, with O_{k-1} the overlap of the two parents' versions one level down, composed with
        the overlap block k already carried -- so that block k+1's creation rotation stays usable for the wavefunction
        transformation (TransformedGuess); costs about two operator rotations. */
    PetscErrorCode RecordBlockBasis(PetscInt k, const std::shared_ptr<dmrgx_host::BasisRotation>& Rnew, int64_t parent_ver, const std::vector<std::vector<EnlPart>>& parts, Block& parent)
    {
        if (k < 0 || k >= (PetscInt)block_rot.size()) return 0;
        RotMeta mnew;
        mnew.parent_ver = parent_ver;
        PetscErrorCode ierr = DescribeEnlarged(parent, parts, mnew); CHKERRQ(ierr);
        const int64_t new_ver = ++ver_counter;
        std::shared_ptr<BasisOverlap> novl;
        const size_t K = (size_t)k;
        const bool child = K + 1 < block_rot.size() && block_rot[K + 1] && rot_meta[K + 1].parent_ver >= 0;
        if (use_guess && use_guess_overlap && child && Rnew && block_rot[K] && HasVectors(*Rnew) && HasVectors(*block_rot[K]) && Rnew->old_sizes == mnew.enl_sizes) {
            const int64_t P = rot_meta[K + 1].parent_ver, old_ver = block_ver[K];
            const BasisOverlap* carried = nullptr;
            bool ok = (P == old_ver);
            if (!ok && block_ovl[K] && block_ovl[K]->from_ver == P && block_ovl[K]->to_ver == old_ver) { carried = block_ovl[K].get(); ok = true; }
            const RotMeta& mold = rot_meta[K];
            std::shared_ptr<dmrgx_host::BasisRotation> RX;
            if (ok) {
                if (mold.parent_ver == parent_ver && mold.enl_sizes == mnew.enl_sizes) RX = block_rot[K];
                else if (K >= 1 && block_ovl[K - 1] && block_ovl[K - 1]->from_ver == mold.parent_ver && block_ovl[K - 1]->to_ver == parent_ver)
                    RX = ProjectRotation(*block_rot[K], mold, *block_ovl[K - 1], mnew);
            }
            if (RX) {
                novl = OverlapOfRotations(*RX, *Rnew, mnew.enl_qn2);
                if (carried) novl = ComposeOverlaps(*carried, *novl);
                novl->from_ver = P; novl->to_ver = new_ver;
            }
        }
        block_ovl[K] = novl;
        block_ver[K] = new_ver;
        rot_meta[K] = std::move(mnew);
        block_rot[K] = Rnew;
        return 0;
    }

    /** Fills `guess` (device, layout of KronBlocks) from the previous step if the two steps are consecutive positions
        of a sweep; returns used = false (guess untouched) otherwise or when any dimension does not line up. */
    PetscErrorCode TransformedGuess(KronBlocks_t& KronBlocks, Block& SysBlock, Block& EnvBlock, const Vec& guess, bool& used, double& min_norm2)
    {
        used = false;
        bool projected = false;
        if (!use_guess || !prev.valid) return 0;
        const PetscInt insys = BlockIndex(SysBlock), inenv = BlockIndex(EnvBlock);
        if (insys < 0 || inenv < 0) return 0;
        const bool grow_left = (insys == prev.outsys && inenv == prev.inenv - 1);
        const bool grow_right = (inenv == prev.outenv && insys == prev.insys - 1);
        if (!grow_left && !grow_right) return 0;
        using dmrgx_host::BasisRotation;
        /* G: this step's truncation on the growing side; S: creation rotation of the shrinking side's previous block */
        const std::shared_ptr<BasisRotation> G = grow_left ? prev.rotL : prev.rotR;
        const PetscInt shrink_idx = grow_left ? prev.inenv : prev.insys;
        if (shrink_idx < 0 || shrink_idx >= (PetscInt)block_rot.size() || !block_rot[(size_t)shrink_idx] || !G) return 0;
        std::shared_ptr<BasisRotation> S = block_rot[(size_t)shrink_idx];
        /* S was computed in a version of block shrink_idx - 1; the block stored there now may be a later one (warm-up) */
        if (shrink_idx >= 1 && (size_t)shrink_idx < rot_meta.size() && rot_meta[(size_t)shrink_idx].parent_ver >= 0 && rot_meta[(size_t)shrink_idx].parent_ver != block_ver[(size_t)shrink_idx - 1]) {
            const RotMeta& mS = rot_meta[(size_t)shrink_idx];
            const std::shared_ptr<BasisOverlap>& ov = block_ovl[(size_t)shrink_idx - 1];
            if (!use_guess_overlap || !ov || ov->from_ver != mS.parent_ver || ov->to_ver != block_ver[(size_t)shrink_idx - 1] || !HasVectors(*S)) return 0;
            RotMeta mC;
            std::vector<std::vector<EnlPart>> parts_now;
            Block& shrink_now = grow_left ? EnvBlock : SysBlock;
            PetscErrorCode ierr0 = EnlargedParts(shrink_now, parts_now); CHKERRQ(ierr0);
            ierr0 = DescribeEnlarged(shrink_now, parts_now, mC); CHKERRQ(ierr0);
            S = ProjectRotation(*S, mS, *ov, mC);
            projected = true;
        }
        Block& Lnew = KronBlocks.LeftBlockRefMod();
        Block& Rnew = KronBlocks.RightBlockRefMod();
        /* new enlarged sectors: growing side = (new block (x) site), shrinking side must equal S's source basis */
        std::vector<std::vector<EnlPart>> parts_grow;
        PetscErrorCode ierr = EnlargedParts(grow_left ? SysBlock : EnvBlock, parts_grow); CHKERRQ(ierr);
        Block& ShrinkEnl = grow_left ? Rnew : Lnew;
        Block& GrowEnl = grow_left ? Lnew : Rnew;
        const std::vector<int32_t> shrink_sizes = ShrinkEnl.Magnetization.Sizes32(), grow_sizes = GrowEnl.Magnetization.Sizes32();
        if (shrink_sizes != S->old_sizes) return 0;
        if (parts_grow.size() != grow_sizes.size()) return 0;
        /* (new growing-block sector a, site state) -> (enlarged sector, offset) */
        std::map<std::pair<int32_t, int32_t>, std::pair<int32_t, int32_t>> where;
        for (size_t I = 0; I < parts_grow.size(); ++I)
            for (const EnlPart& e : parts_grow[I]) where[{e.old_sector, e.site_sector}] = {(int32_t)I, e.off};
        const std::vector<std::vector<EnlPart>>& parts_shrink_prev = grow_left ? prev.partsR : prev.partsL;
        const int32_t nG = (int32_t)G->kept.size(), nS = (int32_t)S->kept.size();
        /* sizes of the growing block's sectors must be G's kept counts, of the shrinking prev block S's kept counts */
        {
            const std::vector<int32_t> gs = (grow_left ? SysBlock : EnvBlock).Magnetization.Sizes32();
            if ((int32_t)gs.size() != nG) return 0;
            for (int32_t a = 0; a < nG; ++a) if (gs[(size_t)a] != G->kept[(size_t)a]) return 0;
        }
        std::vector<std::vector<int32_t>> G_of_old(G->old_sizes.size());
        for (int32_t a = 0; a < nG; ++a) G_of_old[(size_t)G->old_sector[(size_t)a]].push_back(a);

        /* stage 1: Phi = R_G applied on the growing side of every previous KronBlock */
        struct PhiBlock { int32_t a, other, rows, cols; int64_t off; };
        std::vector<PhiBlock> phis;
        std::vector<dmrgx_gemm_task> t1;
        int64_t phi_total = 0;
        /* transposed copies of the rotation blocks (the right-growing case multiplies from the other side): one arena, one
           memset and one batched transpose instead of a buffer, a memset and a launch per sector */
        std::vector<const double*> GT((size_t)nG, nullptr), ST((size_t)nS, nullptr);
        std::shared_ptr<dmrgx_host::DevBuffer> tarena;
        if (!grow_left) {
            std::vector<dmrgx_axpy_task> tr;
            std::vector<int64_t> off;
            int64_t tot = 0;
            auto add = [&](const std::shared_ptr<dmrgx_host::DevBuffer>& rt, int32_t kept, int32_t n) {
                dmrgx_axpy_task k;
                k.dst = nullptr; k.dst_base = nullptr; k.src = rt->dev_ro(); k.ldd = kept; k.lds = n; k.nr = n; k.nc = kept; k.transposed = 1; k.alpha = 1.0;
                tr.push_back(k); off.push_back(tot); tot += (int64_t)kept * n;
            };
            std::vector<int32_t> ia, ij;
            for (int32_t a = 0; a < nG; ++a) if (G->kept[(size_t)a] > 0 && G->rt[(size_t)a]) { add(G->rt[(size_t)a], G->kept[(size_t)a], G->old_sizes[(size_t)G->old_sector[(size_t)a]]); ia.push_back(a); }
            for (int32_t j = 0; j < nS; ++j) if (S->kept[(size_t)j] > 0 && S->rt[(size_t)j] && S->old_sector[(size_t)j] >= 0) { add(S->rt[(size_t)j], S->kept[(size_t)j], S->old_sizes[(size_t)S->old_sector[(size_t)j]]); ij.push_back(j); }
            if (tot > 0) {
                tarena = std::make_shared<dmrgx_host::DevBuffer>((size_t)tot, dmrgx_host::DevBuffer::device_only_t{});
                double* base = tarena->dev_uninitialised();
                if (dmrgx_memset_zero(base, (size_t)tot * sizeof(double), nullptr)) SETERRQ1(mpi_comm, 1, "%s", dmrgx_last_error());
                for (size_t i = 0; i < tr.size(); ++i) tr[i].dst = base + off[i];
                if (dmrgx_cells_axpy((int32_t)tr.size(), tr.data(), nullptr)) SETERRQ1(mpi_comm, 1, "wavefunction transformation: %s", dmrgx_last_error());
                for (size_t i = 0; i < ia.size(); ++i) GT[(size_t)ia[i]] = base + off[i];
                for (size_t i = 0; i < ij.size(); ++i) ST[(size_t)ij[i]] = base + off[ia.size() + i];
            }
        }
        const double* x = prev.psi->buf->dev_ro();
        for (const auto& kb : prev.kb) {
            const int32_t IL = (int32_t)kb[0], IR = (int32_t)kb[1], nL = (int32_t)kb[2], nR = (int32_t)kb[3];
            const int32_t grow_sector = grow_left ? IL : IR;
            if (grow_sector >= (int32_t)G_of_old.size()) return 0;
            for (int32_t a : G_of_old[(size_t)grow_sector]) {
                const int32_t ka = G->kept[(size_t)a];
                if (ka == 0) continue;
                PhiBlock pb;
                pb.a = a; pb.other = grow_left ? IR : IL; pb.off = phi_total;
                if (grow_left) { pb.rows = ka; pb.cols = nR; if (G->old_sizes[(size_t)IL] != nL) return 0; }
                else           { pb.rows = nL; pb.cols = ka; if (G->old_sizes[(size_t)IR] != nR) return 0; }
                phi_total += (int64_t)pb.rows * pb.cols;
                phis.push_back(pb);
                (void)kb;
            }
        }
        if (phis.empty()) return 0;
        auto phi = std::make_shared<dmrgx_host::DevBuffer>((size_t)phi_total, dmrgx_host::DevBuffer::device_only_t{});
        double* ph = phi->dev_uninitialised();
        {
            size_t ip = 0;
            for (const auto& kb : prev.kb) {
                const int32_t IL = (int32_t)kb[0], IR = (int32_t)kb[1], nL = (int32_t)kb[2], nR = (int32_t)kb[3];
                for (int32_t a : G_of_old[(size_t)(grow_left ? IL : IR)]) {
                    const int32_t ka = G->kept[(size_t)a];
                    if (ka == 0) continue;
                    const PhiBlock& pb = phis[ip++];
                    if (!G->rt[(size_t)a]) return 0;            /* a rotation kept for its sector table only */
                    if (grow_left)          /* (ka x nL) . (nL x nR) */
                        t1.push_back(dmrgx_gemm_task{ka, nR, nL, 0, G->rt[(size_t)a]->dev_ro(), nL, x + kb[4], nR, ph + pb.off, nR});
                    else {                  /* (nL x nR) . (nR x ka) */
                        if (!GT[(size_t)a]) return 0;
                        t1.push_back(dmrgx_gemm_task{nL, ka, nR, 0, x + kb[4], nR, GT[(size_t)a], ka, ph + pb.off, ka});
                    }
                }
            }
        }
        /* stage 2: expand the shrinking side through S and place the pieces in the new KronBlocks */
        std::vector<dmrgx_gemm_task> t2;
        double* y = guess->buf->dev_uninitialised();
        for (const PhiBlock& pb : phis) {
            if (pb.other >= (int32_t)parts_shrink_prev.size()) return 0;
            for (const EnlPart& e : parts_shrink_prev[(size_t)pb.other]) {      /* (sector j of the previous shrinking block, site state) */
                const int32_t j = e.old_sector;
                if (j >= nS || S->kept[(size_t)j] != e.size) return 0;
                if (e.size == 0) continue;
                if (S->old_sector[(size_t)j] < 0) continue;                     /* (projected rotation: that sector has no counterpart in the current basis) */
                if (!S->rt[(size_t)j]) return 0;
                const int32_t Jnew = S->old_sector[(size_t)j], nJ = S->old_sizes[(size_t)Jnew];
                auto it = where.find({pb.a, e.site_sector});
                if (it == where.end()) continue;                                /* that (sector, site state) does not exist in the new block */
                const int32_t Inew = it->second.first, goff = it->second.second;
                const PetscInt knew = grow_left ? KronBlocks.Map(Inew, Jnew) : KronBlocks.Map(Jnew, Inew);
                if (knew < 0) continue;
                const int64_t base = KronBlocks.Offsets(knew);
                if (grow_left) {            /* rows = growing part (ka), cols = all of new env sector Jnew */
                    const int32_t ldn = nJ;
                    t2.push_back(dmrgx_gemm_task{pb.rows, nJ, e.size, 0, ph + pb.off + e.off, pb.cols, S->rt[(size_t)j]->dev_ro(), nJ,
                                                 y + base + (int64_t)goff * ldn, ldn});
                } else {                    /* rows = all of new sys sector Jnew, cols = growing part (ka) at column offset goff */
                    if (!ST[(size_t)j]) return 0;
                    const int32_t ldn = grow_sizes[(size_t)Inew];
                    t2.push_back(dmrgx_gemm_task{nJ, pb.cols, e.size, 0, ST[(size_t)j], e.size, ph + pb.off + (int64_t)e.off * pb.cols, pb.cols,
                                                 y + base + goff, ldn});
                }
            }
        }
        if (t2.empty()) return 0;
        if (dmrgx_memset_zero(y, (size_t)guess->n * sizeof(double), nullptr)) SETERRQ1(mpi_comm, 1, "%s", dmrgx_last_error());
        if (dmrgx_dgemm_batch((int32_t)t1.size(), t1.data(), nullptr)) SETERRQ1(mpi_comm, 1, "wavefunction transformation: %s", dmrgx_last_error());
        if (dmrgx_dgemm_batch((int32_t)t2.size(), t2.data(), nullptr)) SETERRQ1(mpi_comm, 1, "wavefunction transformation: %s", dmrgx_last_error());
        /* A projection through basis overlaps drops whatever the current version of the block does not hold (sectors or parts without
           a matching overlap cell are left zero): the previous state had norm 1, so the norm of the result says how much survived.
           Below one half (norm^2 < 0.25) the vector is not a start vector any more -- at zero it would "converge" at E = 0 in a solver
           that trusts it.  The solver checks it (dmrgx_eigs_opts.min_initial_norm2) with the first coefficients it reads anyway and
           falls back to the random vector; round 4's first version took the norm here, a synchronisation of its own in front of every
           projected step's solve. */
        min_norm2 = projected ? 0.25 : 0.0;
        used = true;
        if (projected) ++guesses_projected;
        return 0;
    }

    /* ---- checkpoint / restart (reference include/DMRGBlockContainer.hpp:2456-2481,2689-2764) ---------- */
    static std::string BlockDir(const std::string& BlockType, const PetscInt& iblock) { char b[64]; snprintf(b, sizeof(b), "%s_%09lld/", BlockType.c_str(), LLD(iblock)); return b; }
    static std::string SweepDir(const PetscInt& isweep) { char b[64]; snprintf(b, sizeof(b), "Sweep_%09lld/", LLD(isweep)); return b; }

    /** "-key value" lines of a file go into the options database (PetscOptionsInsertFile of the reference). */
    PetscErrorCode SetOptionsFromFile(const std::string& filename)
    {
        std::ifstream f(filename);
        if (!f) SETERRQ1(mpi_comm, PETSC_ERR_FILE_OPEN, "cannot read %s", filename.c_str());
        std::string line;
        while (std::getline(f, line)) {
            std::istringstream ls(line);
            std::string key, val;
            if (!(ls >> key) || key.size() < 2 || key[0] != '-') continue;
            ls >> val;
            PetscErrorCode ierr = PetscOptionsSetValue(NULL, key.c_str(), val.c_str()); CHKERRQ(ierr);
        }
        return 0;
    }
    PetscErrorCode RetrieveInfoFile(const std::string& filename, std::map<std::string, PetscInt>& dict)
    {
        std::ifstream f(filename);
        if (!f) SETERRQ1(mpi_comm, PETSC_ERR_FILE_OPEN, "cannot read %s", filename.c_str());
        std::string key; long long val;
        while (f >> key >> val) dict[key] = (PetscInt)val;
        return 0;
    }
    /** the sweep-schedule options, as given on the command line (reference :2689-2722) */
    PetscErrorCode SaveAsOptions(const std::string& filename)
    {
        std::ofstream f(filename);
        if (!f) SETERRQ1(mpi_comm, PETSC_ERR_FILE_OPEN, "cannot write %s", filename.c_str());
        char val[4096]; PetscBool set;
        for (const char* key : {"-spin", "-mstates", "-mwarmup", "-nsweeps", "-msweeps", "-maxnsweeps", "-min_block"}) {
            PetscErrorCode ierr = PetscOptionsGetString(NULL, NULL, key, val, sizeof(val), &set); CHKERRQ(ierr);
            if (set) f << key << " " << (val[0] ? val : "yes") << "\n";
        }
        return 0;
    }
    /** End of the warm-up / of a sweep: everything a later run needs to continue from here -- the model
        (Hamiltonian.dat), the sweep schedule (PetscOptions.dat), the counters (Sweep.dat) and the first num_sites/2
        blocks (Sys_%09d/), under scratch_dir/Sweep_%09d/ (reference :2724-2764 + Block::SaveAndDestroy). */
    PetscErrorCode SaveSweepsData()
    {
        if (!do_scratch_dir || mpi_rank) return 0;
        PetscErrorCode ierr;
        const std::string dir = scratch_dir + SweepDir(LoopIdx);
        ierr = Makedir(dir); CHKERRQ(ierr);
        for (PetscInt ib = 0; ib < sys_ninit; ++ib) {
            ierr = Makedir(dir + BlockDir("Sys", ib)); CHKERRQ(ierr);
            ierr = sys_blocks[(size_t)ib].SaveToDisk(dir + BlockDir("Sys", ib)); CHKERRQ(ierr);
        }
        ierr = Ham.SaveAsOptions(dir + "Hamiltonian.dat"); CHKERRQ(ierr);
        ierr = SaveAsOptions(dir + "PetscOptions.dat"); CHKERRQ(ierr);
        std::ofstream f(dir + "Sweep.dat");          /* written last: its presence marks the checkpoint as complete */
        if (!f) SETERRQ1(mpi_comm, PETSC_ERR_FILE_OPEN, "cannot write %sSweep.dat", dir.c_str());
        const PetscInt num_env_blocks = 1, env_ninit = 0;
        auto dump = [&](const char* k, PetscInt v) { f << std::setw(20) << k << "  " << v << "\n"; };
        dump("GlobIdx", GlobIdx); dump("LoopIdx", LoopIdx); dump("num_sys_blocks", num_sys_blocks); dump("num_env_blocks", num_env_blocks);
        dump("sys_ninit", sys_ninit); dump("env_ninit", env_ninit); dump("num_sites", num_sites); dump("sweep_mode", (PetscInt)sweep_mode); dump("msweep_idx", msweep_idx);
        return 0;
    }

    /** Correlations.json: {"info": [...one entry per correlator...], "values": [[...one row per measurement...]]}
        (schema of include/DMRGBlockContainer.hpp:2071-2097,2318-2335 of the reference; values are written with 15
        significant digits instead of the reference's %g). */
    PetscErrorCode PrintCorrelationHeaders()
    {
        if (mpi_rank || !fp_corr || corr_headers_printed) return 0;
        fprintf(fp_corr, "{\n  \"info\" :\n  [\n");
        for (size_t i = 0; i < measurements.size(); ++i) {
            const Correlator& c = measurements[i];
            fprintf(fp_corr, "%s    {\n      \"corrIdx\" : %lld,\n      \"name\"    : \"%s\",\n      \"desc1\"   : \"%s\",\n      \"desc2\"   : \"%s\",\n      \"desc3\"   : \"%s\"\n    }",
                    i ? ",\n" : "", LLD(c.idx), c.name.c_str(), c.desc1.c_str(), c.desc2.c_str(), c.desc3.c_str());
        }
        fprintf(fp_corr, "\n  ],\n  \"values\" :\n  [\n");
        fflush(fp_corr);
        corr_headers_printed = PETSC_TRUE;
        return 0;
    }

    /** Product of the listed single-site operators of one block, in the block's basis (the reference multiplies the
        retrieved matrices with MatMatMult, include/DMRGBlockContainer.hpp:2333-2410).  A single operator is handed out
        as it is stored (views included), an empty list is the identity; products are formed on the device, one MFMA
        GEMM per sector block. */
    typedef std::map<std::pair<int, PetscInt>, Mat> DenseCache;        /**< (operator type, site) -> dense form, per measurement */
    PetscErrorCode CalculateOperatorProduct(Block& blk, const std::vector<Op>& ops, Mat& out, DenseCache& cache, const bool want_dense = false)
    {
        PetscErrorCode ierr;
        auto fetch = [&](const Op& o, Mat& m) -> PetscErrorCode {
            if (o.idx < 0 || o.idx >= blk.NumSites()) SETERRQ2(mpi_comm, PETSC_ERR_ARG_OUTOFRANGE, "Correlator site %lld outside the block's %lld sites.", LLD(o.idx), LLD(blk.NumSites()));
            switch (o.OpType) {
                case OpSz: m = blk.Sz(o.idx); break;
                case OpSp: m = blk.Sp(o.idx); break;
                case OpSm: m = blk.Sm(o.idx); break;
                default: SETERRQ(mpi_comm, PETSC_ERR_ARG_WRONG, "Correlators take Sz, Sp and Sm operators.");
            }
            if (!m) SETERRQ2(mpi_comm, PETSC_ERR_ARG_WRONGSTATE, "Correlator operator %s(%lld) is not resident in the block (pruned): register correlators before Warmup().", OpToCStr(o.OpType), LLD(o.idx));
            return 0;
        };
        auto dense = [&](const Op& o, Mat& d) -> PetscErrorCode {
            const auto key = std::make_pair((int)o.OpType, o.idx);
            auto it = cache.find(key);
            if (it != cache.end()) { d = it->second; return 0; }
            Mat m;
            PetscErrorCode e = fetch(o, m); CHKERRQ(e);
            e = dmrgx_host::SectorMatDensify(m, d); if (e) SETERRQ1(mpi_comm, 1, "operator product: %s", dmrgx_last_error());
            cache[key] = d;
            return 0;
        };
        if (ops.empty()) {                                          /* identity: one scaled-identity cell per sector */
            out = std::make_shared<dmrgx_host::SectorMat>();
            out->shift = 0; out->sizes = blk.Magnetization.Sizes32();
            for (int32_t q = 0; q < (int32_t)out->sizes.size(); ++q) {
                dmrgx_host::MatCell c;
                c.q = q; c.nr = c.nc = out->sizes[q]; c.kind = DMRGX_CELL_IDENT; c.scale = 1.0;
                out->cells.push_back(c);
            }
            return 0;
        }
        if (ops.size() == 1 && !want_dense) return fetch(ops[0], out);
        Mat prod;
        ierr = dense(ops[0], prod); CHKERRQ(ierr);
        for (size_t i = 1; i < ops.size(); ++i) {
            Mat next, tmp;
            ierr = dense(ops[i], next); CHKERRQ(ierr);
            ierr = dmrgx_host::SectorMatMatMult(prod, next, tmp); if (ierr) SETERRQ1(mpi_comm, 1, "operator product: %s", dmrgx_last_error());
            prod = tmp;
        }
        out = prod;
        return 0;
    }

    /** < psi | (product of SysOps) (x) (product of EnvOps) | psi > for every registered correlator, through a one-term
        superblock plan, one MatMult and one dot product each (include/DMRGBlockContainer.hpp:2255-2303). */
    PetscErrorCode CalculateCorrelations_BlockDiag(KronBlocks_t& KronBlocks, const Vec& gsv_r, const PetscBool flg = PETSC_TRUE)
    {
        PetscErrorCode ierr = PrintCorrelationHeaders(); CHKERRQ(ierr);
        if (!flg) return 0;
        std::vector<PetscScalar> CorrValues(measurements.size(), 0.0);
        Block& L = KronBlocks.LeftBlockRefMod();
        Block& R = KronBlocks.RightBlockRefMod();
        bool need_sm = false;
        for (const Correlator& c : measurements) { for (const Op& o : c.SysOps) need_sm |= (o.OpType == OpSm); for (const Op& o : c.EnvOps) need_sm |= (o.OpType == OpSm); }
        const bool l_had = L.HasSm(), r_had = R.HasSm();
        if (need_sm && !l_had) { ierr = L.CreateSm(); CHKERRQ(ierr); }
        if (need_sm && !R.HasSm()) { ierr = R.CreateSm(); CHKERRQ(ierr); }
        Vec Op_Vec;
        {
            Op_Vec = std::make_shared<dmrgx_host::VecImpl>();
            Op_Vec->n = gsv_r->n;
            Op_Vec->buf = std::make_shared<dmrgx_host::DevBuffer>((size_t)gsv_r->n, dmrgx_host::DevBuffer::device_only_t{});
        }
        DenseCache cacheL, cacheR;
        const PetscInt nkb = KronBlocks.size();
        dmrgx_host::DevBuffer dev_vals(std::max<size_t>(measurements.size(), 1), dmrgx_host::DevBuffer::device_only_t{});
        if (dmrgx_memset_zero(dev_vals.dev_uninitialised(), std::max<size_t>(measurements.size(), 1) * sizeof(double), nullptr)) SETERRQ1(mpi_comm, 1, "%s", dmrgx_last_error());
        std::vector<char> queued(measurements.size(), 0);
        BuildNeedTables();
        dmrgx_comm* corr_comm = dmrgx_host::WorldComm();
        const int corr_me = corr_comm ? dmrgx_host::WorldRank() : 0;
        auto mine = [&](size_t ic) { return corr_owner.size() != measurements.size() || corr_owner[ic] < 0 || corr_owner[ic] == corr_me; };
        PetscLogDouble tc0, tc1, t_one = 0, t_two = 0, t_batch = 0; PetscInt n_one = 0, n_two = 0, n_batch = 0;
        /* ---- system-block correlators of one or two operators (magnetisations, neighbour pairs: the bulk of the table), batched:
                <psi| P (x) 1 |psi> = sum_k < P[IL(k)], X_k X_k^T >_F
           so the Gram blocks G_k = X_k X_k^T are formed once (one grouped GEMM), all operator pairs are multiplied in one
           grouped GEMM per chunk, and all expectation values are one batch of 2-D inner products. */
        std::vector<char> done(measurements.size(), 0);
        for (size_t ic = 0; ic < measurements.size(); ++ic) if (!mine(ic)) done[ic] = 1;      /* another rank's: its value arrives with the all-reduce */
        if (use_corr_batch) {
            PetscTime(&tc0);
            std::vector<int64_t> g_off((size_t)nkb + 1, 0);
            for (PetscInt k = 0; k < nkb; ++k) { const int64_t nl = L.Magnetization.Sizes(KronBlocks.LeftIdx(k)); g_off[(size_t)k + 1] = g_off[(size_t)k] + nl * nl; }
            dmrgx_host::DevBuffer xt((size_t)std::max<PetscInt>(gsv_r->n, 1), dmrgx_host::DevBuffer::device_only_t{});
            dmrgx_host::DevBuffer gram((size_t)std::max<int64_t>(g_off[(size_t)nkb], 1), dmrgx_host::DevBuffer::device_only_t{});
            {
                if (dmrgx_memset_zero(xt.dev_uninitialised(), (size_t)gsv_r->n * sizeof(double), nullptr)) SETERRQ1(mpi_comm, 1, "%s", dmrgx_last_error());
                std::vector<dmrgx_axpy_task> tr; std::vector<dmrgx_gemm_task> gt;
                const double* x = gsv_r->buf->dev_ro();
                for (PetscInt k = 0; k < nkb; ++k) {
                    const int32_t nl = (int32_t)L.Magnetization.Sizes(KronBlocks.LeftIdx(k)), nr = (int32_t)R.Magnetization.Sizes(KronBlocks.RightIdx(k));
                    if (nl == 0 || nr == 0) continue;
                    double* xtk = xt.dev_uninitialised() + KronBlocks.Offsets(k);
                    dmrgx_axpy_task t; t.dst = xtk; t.dst_base = nullptr; t.src = x + KronBlocks.Offsets(k); t.ldd = nl; t.lds = nr; t.nr = nr; t.nc = nl; t.transposed = 1; t.alpha = 1.0;
                    tr.push_back(t);
                    gt.push_back(dmrgx_gemm_task{nl, nl, nr, 0, x + KronBlocks.Offsets(k), nr, xtk, nl, gram.dev_uninitialised() + g_off[(size_t)k], nl});
                }
                if (!tr.empty() && dmrgx_cells_axpy((int32_t)tr.size(), tr.data(), nullptr)) SETERRQ1(mpi_comm, 1, "dmrgx_cells_axpy: %s", dmrgx_last_error());
                if (!gt.empty() && dmrgx_dgemm_batch((int32_t)gt.size(), gt.data(), nullptr)) SETERRQ1(mpi_comm, 1, "dmrgx_dgemm_batch: %s", dmrgx_last_error());
            }
            const std::vector<int32_t> lsz = L.Magnetization.Sizes32();
            const int32_t ns = (int32_t)lsz.size();
            auto cell_at = [](const Mat& M, int32_t q) -> const dmrgx_host::MatCell* { for (const dmrgx_host::MatCell& c : M->cells) if (c.q == q) return &c; return nullptr; };
            std::vector<dmrgx_gemm_task> ptasks;
            std::vector<dmrgx_dot2d_task> dtasks;
            std::vector<std::shared_ptr<dmrgx_host::DevBuffer>> arenas;      /* pair products of the current chunk */
            int64_t chunk_elems = 0;
            const int64_t chunk_limit = (int64_t)1 << 28;                     /* 2 GiB of products per chunk */
            auto flush = [&]() -> PetscErrorCode {
                if (!ptasks.empty() && dmrgx_dgemm_batch((int32_t)ptasks.size(), ptasks.data(), nullptr)) SETERRQ1(mpi_comm, 1, "dmrgx_dgemm_batch: %s", dmrgx_last_error());
                if (!dtasks.empty() && dmrgx_dot2d_batch((int32_t)dtasks.size(), dtasks.data(), dev_vals.dev_uninitialised(), nullptr)) SETERRQ1(mpi_comm, 1, "dmrgx_dot2d_batch: %s", dmrgx_last_error());
                ptasks.clear(); dtasks.clear(); arenas.clear(); chunk_elems = 0;
                return 0;
            };
            for (size_t ic = 0; ic < measurements.size(); ++ic) {
                const Correlator& c = measurements[ic];
                if (done[ic]) continue;
                if (!c.EnvOps.empty() || c.SysOps.empty() || c.SysOps.size() > 2) continue;
                int shift = 0;
                for (const Op& o : c.SysOps) shift += int(o.OpType);
                if (shift != 0) continue;                                       /* left to the general loop: it records the zero */
                Mat A, B;
                ierr = CalculateOperatorProduct(L, {c.SysOps[0]}, A, cacheL, true); CHKERRQ(ierr);
                if (c.SysOps.size() == 2) { ierr = CalculateOperatorProduct(L, {c.SysOps[1]}, B, cacheL, true); CHKERRQ(ierr); }
                bool any = false;
                std::shared_ptr<dmrgx_host::DevBuffer> arena;
                int64_t cursor = 0;
                if (B) {
                    int64_t total = 0;
                    for (PetscInt k = 0; k < nkb; ++k) { const int64_t nl = lsz[(size_t)KronBlocks.LeftIdx(k)]; total += nl * nl; }
                    if (chunk_elems + total > chunk_limit) { ierr = flush(); CHKERRQ(ierr); }
                    arena = std::make_shared<dmrgx_host::DevBuffer>((size_t)std::max<int64_t>(total, 1), dmrgx_host::DevBuffer::device_only_t{});
                    arenas.push_back(arena);
                    chunk_elems += total;
                }
                for (PetscInt k = 0; k < nkb; ++k) {
                    const int32_t q = (int32_t)KronBlocks.LeftIdx(k), nl = lsz[(size_t)q];
                    if (nl == 0 || R.Magnetization.Sizes(KronBlocks.RightIdx(k)) == 0) continue;
                    const double* gk = gram.dev_ro() + g_off[(size_t)k];
                    if (!B) {
                        const dmrgx_host::MatCell* a = cell_at(A, q);
                        if (!a || A->shift != 0) continue;
                        dtasks.push_back(dmrgx_dot2d_task{a->buf->dev_ro() + a->off, a->ld, gk, nl, nl, nl, (int32_t)c.idx, 0});
                        any = true;
                    } else {
                        const int32_t qa = q + A->shift;                    /* (A B)[q -> q] = A[q -> qa] B[qa -> q] */
                        if (qa < 0 || qa >= ns) continue;
                        const dmrgx_host::MatCell* a = cell_at(A, q);
                        const dmrgx_host::MatCell* b = cell_at(B, qa);
                        if (!a || !b || a->nc == 0) continue;
                        double* pc = arena->dev_uninitialised() + cursor;
                        cursor += (int64_t)nl * nl;
                        ptasks.push_back(dmrgx_gemm_task{nl, nl, a->nc, 0, a->buf->dev_ro() + a->off, a->ld, b->buf->dev_ro() + b->off, b->ld, pc, nl});
                        dtasks.push_back(dmrgx_dot2d_task{pc, nl, gk, nl, nl, nl, (int32_t)c.idx, 0});
                        any = true;
                    }
                }
                done[ic] = 1;
                if (any) queued[(size_t)c.idx] = 1; else CorrValues[c.idx] = 0.0;
                ++n_batch;
            }
            ierr = flush(); CHKERRQ(ierr);
            PetscTime(&tc1);
            t_batch = tc1 - tc0;
        }
        for (size_t ic = 0; ic < measurements.size(); ++ic) {
            const Correlator& c = measurements[ic];
            if (done[ic]) continue;
            PetscTime(&tc0);
            int shift = 0;
            for (const Op& o : c.SysOps) shift += int(o.OpType);
            for (const Op& o : c.EnvOps) shift += int(o.OpType);
            if (shift != 0) { CorrValues[c.idx] = 0.0; continue; }          /* changes the total Sz: no overlap with the target sector */
            if (c.EnvOps.empty()) {
                /* operators on the system block only: (P (x) 1) psi is Y_k = P[IL(k)] X_k for every KronBlock -- one grouped
                   MFMA launch over all blocks, no plan (the bulk of the driver's correlators) */
                Mat P;
                ierr = CalculateOperatorProduct(L, c.SysOps, P, cacheL, true); CHKERRQ(ierr);
                if (dmrgx_memset_zero(Op_Vec->buf->dev_uninitialised(), (size_t)gsv_r->n * sizeof(double), nullptr)) SETERRQ1(mpi_comm, 1, "%s", dmrgx_last_error());
                std::vector<dmrgx_gemm_task> tasks;
                const double* x = gsv_r->buf->dev_ro();
                double* y = Op_Vec->buf->dev_uninitialised();
                for (PetscInt k = 0; k < nkb; ++k) {
                    const int32_t il = (int32_t)KronBlocks.LeftIdx(k), ir = (int32_t)KronBlocks.RightIdx(k);
                    const int32_t nl = (int32_t)L.Magnetization.Sizes(il), nr = (int32_t)R.Magnetization.Sizes(ir);
                    for (const dmrgx_host::MatCell& pc : P->cells) {
                        if (pc.q != il) continue;
                        tasks.push_back(dmrgx_gemm_task{nl, nr, nl, 0, pc.buf->dev_ro() + pc.off, pc.ld, x + KronBlocks.Offsets(k), nr, y + KronBlocks.Offsets(k), nr});
                    }
                }
                if (!tasks.empty() && dmrgx_dgemm_batch((int32_t)tasks.size(), tasks.data(), nullptr)) SETERRQ1(mpi_comm, 1, "dmrgx_dgemm_batch: %s", dmrgx_last_error());
            } else {
                Mat PL, PR, KronOp;
                ierr = CalculateOperatorProduct(L, c.SysOps, PL, cacheL); CHKERRQ(ierr);
                ierr = CalculateOperatorProduct(R, c.EnvOps, PR, cacheR); CHKERRQ(ierr);
                ierr = KronBlocks.KronConstructShifted(PL, PR, KronOp); CHKERRQ(ierr);
                ierr = MatMult(KronOp, gsv_r, Op_Vec); CHKERRQ(ierr);
                ierr = MatDestroy_KronSumShell(&KronOp); CHKERRQ(ierr);
                ierr = MatDestroy(&KronOp); CHKERRQ(ierr);
            }
            /* <psi|O|psi> is queued into a device array; all values come back with one copy after the loop */
            if (dmrgx_dot_async(gsv_r->n, Op_Vec->buf->dev_ro(), gsv_r->buf->dev_ro(), dev_vals.dev_uninitialised() + c.idx, nullptr)) SETERRQ1(mpi_comm, 1, "dmrgx_dot_async: %s", dmrgx_last_error());
            queued[(size_t)c.idx] = 1;
            PetscTime(&tc1);
            if (c.EnvOps.empty()) { t_one += tc1 - tc0; ++n_one; } else { t_two += tc1 - tc0; ++n_two; }
        }
        {
            std::vector<double> hv(measurements.size(), 0.0);
            const bool dealt = corr_comm && corr_owner.size() == measurements.size() && !measurements.empty() && corr_owner[0] >= 0;
            if (dealt && dmrgx_comm_allreduce_sum(corr_comm, dev_vals.dev_uninitialised(), (int64_t)measurements.size(), nullptr)) SETERRQ1(mpi_comm, 1, "dmrgx_comm_allreduce_sum: %s", dmrgx_last_error());
            if (!hv.empty() && dmrgx_memcpy_d2h(hv.data(), dev_vals.dev_ro(), hv.size() * sizeof(double), nullptr)) SETERRQ1(mpi_comm, 1, "%s", dmrgx_last_error());
            for (size_t i = 0; i < hv.size(); ++i) if (queued[i] || dealt) CorrValues[i] = hv[i];      /* (an entry nobody queued stayed zero) */
        }
        if (!mpi_rank && verbose) printf("  * Calc. of Correlators: %lld batched through the Gram blocks %.6f s, %lld more on the system block %.6f s, %lld across the cut %.6f s\n", LLD(n_batch), t_batch, LLD(n_one), t_one, LLD(n_two), t_two);
        if (need_sm && !l_had) { ierr = L.DestroySm(); CHKERRQ(ierr); }
        if (need_sm && !r_had && R.HasSm()) { ierr = R.DestroySm(); CHKERRQ(ierr); }
        if (!mpi_rank && fp_corr) {
            fprintf(fp_corr, "%s    [", corr_printed_first ? ",\n" : "");
            corr_printed_first = PETSC_TRUE;
            for (size_t i = 0; i < measurements.size(); ++i) fprintf(fp_corr, "%s %.15g", i ? "," : "", CorrValues[i]);
            fprintf(fp_corr, " ]");
            fflush(fp_corr);
        }
        return 0;
    }

    /** Reduced density matrices of both sides, their full spectra (device), the global cut to MStates states and the
        rotation matrices.  The ordering rules are the reference's: concatenate the spectra KronBlock by KronBlock,
        stable-sort by decreasing eigenvalue, keep the first min(MStates, #), stable-sort the survivors by sector. */
    /*  need_vec_L / need_vec_R = false (engine extension): that side's rotation is not wanted (its output block is never read
        again).  Its spectrum is then taken from the OTHER side of the same KronBlock -- Psi Psi^T and Psi^T Psi share their
        non-zero eigenvalues, the rest are exact zeros -- so the density matrices of that side are neither built nor
        diagonalised; truncation error and kept-sector table follow from the same sort / cut rules as ever. */
    PetscErrorCode GetTruncation(const KronBlocks_t& KronBlocks, const Vec& gsv_r, const PetscInt& MStates, BasisTransformation& BT_L, BasisTransformation& BT_R,
                                 const PetscInt keyL = -1, const PetscInt keyR = -1, const bool need_vec_L = true, const bool need_vec_R = true)
    {
        PetscErrorCode ierr;
        if (gsv_r->n != KronBlocks.NumStates()) SETERRQ2(PETSC_COMM_SELF, 1, "Incorrect vector length. Expected %lld. Got %lld.", LLD(KronBlocks.NumStates()), LLD(gsv_r->n));
        const QuantumNumbers* M[2] = {&KronBlocks.LeftBlockRef().Magnetization, &KronBlocks.RightBlockRef().Magnetization};
        const std::vector<int32_t> ls = M[0]->Sizes32(), rs = M[1]->Sizes32();
        const PetscInt nb = KronBlocks.size();
        std::vector<int32_t> bil, bir;
        for (PetscInt k = 0; k < nb; ++k) {
            bil.push_back((int32_t)KronBlocks.LeftIdx(k)); bir.push_back((int32_t)KronBlocks.RightIdx(k));
            if (KronBlocks.Offsets(k + 1) - KronBlocks.Offsets(k) != (PetscInt)ls[bil[k]] * rs[bir[k]]) SETERRQ(PETSC_COMM_SELF, 1, "Incorrect segment length.");
        }
        dmrgx_sectors sl{(int32_t)ls.size(), ls.data()}, sr{(int32_t)rs.size(), rs.data()};
        dmrgx_rdm* rdm = nullptr;
        PetscLogDouble tr0, tr1;
        PetscTime(&tr0);
        /* warm start of the block-Jacobi eigensolver: the eigenbasis found at the previous visit of the block being created
           (same index, same sector sizes) almost diagonalises the new density matrix once the sweeps settle */
        const PetscInt keys[2] = {keyL, keyR};
        std::vector<const double*> v0((size_t)(2 * nb), nullptr);
        PetscInt nwarm = 0;
        if (use_rdm_warm && !dmrgx_host::WorldComm()) for (int side = 0; side < 2; ++side) {
            auto it = rdm_basis.find({keys[side], (keyR == keyL) ? 0 : side});      /* centre step: both sides are the same block */
            if (keys[side] < 0 || it == rdm_basis.end() || it->second.sizes != (side == 0 ? ls : rs)) continue;
            for (PetscInt k = 0; k < nb; ++k) {
                auto e = it->second.E.find(side == 0 ? bil[k] : bir[k]);
                if (e != it->second.E.end() && e->second) { v0[(size_t)(2 * k + side)] = e->second->dev_ro(); ++nwarm; }
            }
        }
        /* Multi-GPU (the reference solves every density matrix on rank 0 and broadcasts the rotation,
           include/DMRGBlockContainer.hpp:1673-1677, 1812-1925): the 2 nb density matrices are dealt over the ranks by their n^3
           cost, heaviest first to the least loaded rank (the same deterministic deal on every rank); each rank builds and
           diagonalises its share, the spectra are exchanged, every rank performs the same global sort / m-cut, and the owner
           of a matrix broadcasts its kept eigenvectors. */
        dmrgx_comm* comm = dmrgx_host::WorldComm();
        const int W = comm ? dmrgx_host::WorldSize() : 1, me = comm ? dmrgx_host::WorldRank() : 0;
        const bool need_vec[2] = {need_vec_L || !need_vec_R, need_vec_R};        /* at least one side is solved */
        const bool partial = !need_vec[0] || !need_vec[1];
        std::vector<int> owner((size_t)(2 * nb), 0);                              /* -1: not solved by anyone (spectrum borrowed) */
        for (PetscInt k = 0; k < nb; ++k) for (int side = 0; side < 2; ++side) if (!need_vec[side]) owner[(size_t)(2 * k + side)] = -1;
        if (W > 1) {
            std::vector<std::pair<double, int>> units;
            for (PetscInt k = 0; k < nb; ++k) for (int side = 0; side < 2; ++side) {
                if (!need_vec[side]) continue;
                const double n = side == 0 ? ls[bil[k]] : rs[bir[k]]; units.push_back({n * n * n, (int)(2 * k + side)});
            }
            std::stable_sort(units.begin(), units.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first > b.first; });
            std::vector<double> load((size_t)W, 0.0);
            for (const auto& u : units) { int best = 0; for (int w = 1; w < W; ++w) if (load[(size_t)w] < load[(size_t)best]) best = w; load[(size_t)best] += u.first; owner[(size_t)u.second] = best; }
        }
        if (W > 1 || partial) {
            std::vector<uint8_t> mask((size_t)nb, 0);
            for (PetscInt k = 0; k < nb; ++k) mask[(size_t)k] = (uint8_t)((owner[(size_t)(2 * k)] == me ? 1 : 0) | (owner[(size_t)(2 * k + 1)] == me ? 2 : 0));
            if (dmrgx_rdm_create_subset(&sl, &sr, (int32_t)nb, bil.data(), bir.data(), gsv_r->buf->dev_ro(), mask.data(), nullptr, &rdm))
                SETERRQ1(mpi_comm, 1, "dmrgx_rdm_create_subset: %s", dmrgx_last_error());
        }
        else if (dmrgx_rdm_create_warm(&sl, &sr, (int32_t)nb, bil.data(), bir.data(), gsv_r->buf->dev_ro(), nwarm ? v0.data() : nullptr, nullptr, &rdm))
            SETERRQ1(mpi_comm, 1, "dmrgx_rdm_create: %s", dmrgx_last_error());
        /* spectra of every matrix on every rank */
        std::vector<int64_t> spec_off((size_t)(2 * nb + 1), 0);
        for (PetscInt k = 0; k < nb; ++k) { spec_off[(size_t)(2 * k + 1)] = spec_off[(size_t)(2 * k)] + ls[bil[k]]; spec_off[(size_t)(2 * k + 2)] = spec_off[(size_t)(2 * k + 1)] + rs[bir[k]]; }
        std::vector<double> spectra((size_t)spec_off[(size_t)(2 * nb)], 0.0);
        for (PetscInt k = 0; k < nb; ++k) for (int side = 0; side < 2; ++side) {
            if (owner[(size_t)(2 * k + side)] != me) continue;
            if (dmrgx_rdm_eigenvalues(rdm, side, (int32_t)k, spectra.data() + spec_off[(size_t)(2 * k + side)])) { dmrgx_rdm_destroy(rdm); SETERRQ1(mpi_comm, 1, "dmrgx_rdm_eigenvalues: %s", dmrgx_last_error()); }
        }
        if (W > 1) {
            std::vector<double> all(spectra.size() * (size_t)W);
            if (dmrgx_comm_allgather_host(comm, spectra.data(), all.data(), spectra.size() * sizeof(double), nullptr)) { dmrgx_rdm_destroy(rdm); SETERRQ1(mpi_comm, 1, "dmrgx_comm_allgather_host: %s", dmrgx_last_error()); }
            for (PetscInt u = 0; u < 2 * nb; ++u) {
                if (owner[(size_t)u] < 0) continue;
                std::copy(all.begin() + (int64_t)owner[(size_t)u] * (int64_t)spectra.size() + spec_off[(size_t)u], all.begin() + (int64_t)owner[(size_t)u] * (int64_t)spectra.size() + spec_off[(size_t)u + 1],
                          spectra.begin() + spec_off[(size_t)u]);
            }
        }
        if (partial) for (PetscInt k = 0; k < nb; ++k) {       /* borrowed spectra: the other side's eigenvalues, zero-padded / cut to this side's size */
            const int dead = need_vec[0] ? 1 : 0, live = 1 - dead;
            const int64_t nd = spec_off[(size_t)(2 * k + dead + 1)] - spec_off[(size_t)(2 * k + dead)], nl = spec_off[(size_t)(2 * k + live + 1)] - spec_off[(size_t)(2 * k + live)];
            for (int64_t e = 0; e < nd; ++e) spectra[(size_t)(spec_off[(size_t)(2 * k + dead)] + e)] = e < nl ? spectra[(size_t)(spec_off[(size_t)(2 * k + live)] + e)] : 0.0;
        }
        if (use_rdm_warm && W == 1) for (int side = 0; side < 2; ++side) {      /* remember this visit's eigenbases (all eigenvectors, as rows) */
            if (keys[side] < 0 || (side == 1 && keyR == keyL)) continue;
            /* a side whose spectrum was borrowed (dead block of a pruned sweep) has no eigenvectors: drop what an earlier visit stored */
            if (!need_vec[side]) { rdm_basis.erase({keys[side], side}); continue; }
            WarmBasis& wb = rdm_basis[{keys[side], side}];
            wb.sizes = side == 0 ? ls : rs;
            wb.E.clear();
            for (PetscInt k = 0; k < nb; ++k) {
                const int32_t sec = side == 0 ? bil[k] : bir[k], n = (side == 0 ? ls : rs)[(size_t)sec];
                if (n == 0) continue;
                auto buf = std::make_shared<dmrgx_host::DevBuffer>((size_t)n * n, dmrgx_host::DevBuffer::device_only_t{});
                if (dmrgx_rdm_eigenvectors(rdm, side, (int32_t)k, n, buf->dev_uninitialised(), n, nullptr)) { dmrgx_rdm_destroy(rdm); SETERRQ1(mpi_comm, 1, "dmrgx_rdm_eigenvectors: %s", dmrgx_last_error()); }
                wb.E[sec] = buf;
            }
        }
        PetscTime(&tr1);
        {
            /* which path the density-matrix solver took: a time-out fallback or an unexpected launch path must show in DMRGRun.json, not only on stderr */
            dmrgx_rdm_report rr;
            if (dmrgx_rdm_info(rdm, &rr)) { dmrgx_rdm_destroy(rdm); SETERRQ1(mpi_comm, 1, "dmrgx_rdm_info: %s", dmrgx_last_error()); }
            ++rdm_calls;
            if (rr.solver == 1) ++rdm_jacobi_calls;
            if (rr.timed_out) ++trid_fallbacks;
            if (rr.trid_launch_matrices > 0) ++trid_launch_calls;
            if (rr.trid_persistent_matrices > 0) ++trid_persistent_calls;
            trid_max_wgs = std::max<PetscInt>(trid_max_wgs, rr.max_workgroups_per_matrix);
            rdm_max_levels = std::max<PetscInt>(rdm_max_levels, rr.merge_levels);
            rdm_max_wy_blocks = std::max<PetscInt>(rdm_max_wy_blocks, rr.wy_blocks_max);
            if (!mpi_rank && verbose) printf("  RDM: %lld KronBlocks, %lld warm-started, block-Jacobi sweeps %d, persistent / launch-path matrices %d / %d, workgroups per matrix <= %d, merge levels %d, create %.6f s\n",
                                             LLD(nb), LLD(nwarm), rr.n_sweeps, rr.trid_persistent_matrices, rr.trid_launch_matrices, rr.max_workgroups_per_matrix, rr.merge_levels, tr1 - tr0);
        }
        BasisTransformation* BT[2] = {&BT_L, &BT_R};
        /* Two phases (round 5): first the m-cut of both sides on the spectra alone, then -- dmrgx_rdm_select -- the eigenvectors of the kept
           states only (the solver's last merge, back-transformation and Rayleigh quotients run on half-width matrices when half of the
           states are kept), then the rotations. */
        std::map<PetscInt, std::pair<PetscInt, PetscInt>> per_side[2];       /* blkIdx -> (KronBlock, count) */
        for (int side = 0; side < 2; ++side) {
            std::vector<Eigen_t> eigen;
            eigen.reserve((size_t)(spec_off[(size_t)(2 * nb)] / 2 + 1));
            std::vector<size_t> run_end;                                   /* ends of the per-KronBlock runs of `eigen` */
            bool runs_sorted = true;
            for (PetscInt k = 0; k < nb; ++k) {
                const PetscInt blk = side == 0 ? bil[k] : bir[k], n = M[side]->Sizes(blk);
                const double* w = spectra.data() + spec_off[(size_t)(2 * k + side)];
                for (PetscInt e = 0; e < n; ++e) { eigen.push_back({w[(size_t)e], k, e, blk}); if (e > 0 && w[(size_t)e] > w[(size_t)e - 1]) runs_sorted = false; }
                run_end.push_back(eigen.size());
            }
            ierr = SaveEntanglementSpectrum(side, eigen, *M[side]); CHKERRQ(ierr);
            /* stable_sort(greater_eigval) of the reference (include/DMRGBlockContainer.hpp:1795).  Every block's spectrum arrives in
               descending order, so the stable sort is a stable merge of nb sorted runs (ties keep the KronBlock order either way):
               log2(nb) merge passes instead of log2(#eigenvalues) -- 0.2 ms of a configs[3] step went into the sort
               (profiles/r04_hostprof_glue_m2048.txt).  Should a run not be sorted, the sort itself is done. */
            auto gt = [](const Eigen_t& a, const Eigen_t& b) { return a.eigval > b.eigval; };
            if (runs_sorted) {
                std::vector<size_t> ends = run_end;
                while (ends.size() > 1) {
                    std::vector<size_t> next;
                    size_t begin = 0;
                    for (size_t r = 0; r < ends.size(); r += 2) {
                        if (r + 1 < ends.size()) { std::inplace_merge(eigen.begin() + (std::ptrdiff_t)begin, eigen.begin() + (std::ptrdiff_t)ends[r], eigen.begin() + (std::ptrdiff_t)ends[r + 1], gt); begin = ends[r + 1]; next.push_back(ends[r + 1]); }
                        else { begin = ends[r]; next.push_back(ends[r]); }
                    }
                    ends.swap(next);
                }
            } else std::stable_sort(eigen.begin(), eigen.end(), gt);
            const PetscInt m = PetscMin(MStates, (PetscInt)eigen.size());
            eigen.resize((size_t)m);
            std::stable_sort(eigen.begin(), eigen.end(), [](const Eigen_t& a, const Eigen_t& b) { return a.blkIdx < b.blkIdx; });   /* stable_sort(less_blkIdx), :1852 */
            PetscReal trunc = 1.0;
            for (const Eigen_t& e : eigen) trunc -= (e.eigval > 0) * e.eigval;
            BT[side]->TruncErr = trunc;
            /* kept states per sector are the top ones of that sector's spectrum, in decreasing order */
            std::map<PetscInt, std::pair<PetscInt, PetscInt>>& per = per_side[side];
            for (const Eigen_t& e : eigen) { auto& p = per[e.blkIdx]; p.first = e.seqIdx; p.second += 1; }
        }
        {
            std::vector<int32_t> counts((size_t)(2 * nb), 0);
            for (int side = 0; side < 2; ++side) for (const auto& kv : per_side[side]) counts[(size_t)(2 * kv.second.first + side)] = (int32_t)kv.second.second;
            if (dmrgx_rdm_select(rdm, counts.data(), nullptr)) { dmrgx_rdm_destroy(rdm); SETERRQ1(mpi_comm, 1, "dmrgx_rdm_select: %s", dmrgx_last_error()); }
        }
        for (int side = 0; side < 2; ++side) {
            const std::map<PetscInt, std::pair<PetscInt, PetscInt>>& per = per_side[side];
            auto rot = std::make_shared<dmrgx_host::BasisRotation>();
            rot->old_sizes = M[side]->Sizes32();
            std::vector<PetscReal> qn_list; std::vector<PetscInt> qn_size;
            /* On several ranks the kept eigenvectors travel in ONE broadcast per owner and side (round 2: one per kept sector): the owner
               gathers the rows of all its sectors into a staging buffer, broadcasts it, and every rank cuts its per-sector buffers out
               of it with device copies (the reference broadcasts the whole rotation from rank 0, include/DMRGBlockContainer.hpp:1812-1925). */
            std::vector<int64_t> stage_off;                              /* per kept sector: offset into its owner's staging buffer */
            std::vector<int64_t> stage_len((size_t)W, 0);
            for (const auto& kv : per) {
                const PetscInt blk = kv.first, k = kv.second.first, cnt = kv.second.second, n = M[side]->Sizes(blk);
                const int own = owner[(size_t)(2 * k + side)];
                stage_off.push_back(own >= 0 ? stage_len[(size_t)own] : -1);
                if (own >= 0) stage_len[(size_t)own] += (int64_t)cnt * n;
            }
            std::vector<std::shared_ptr<dmrgx_host::DevBuffer>> staging((size_t)W);
            if (W > 1) for (int w = 0; w < W; ++w) if (stage_len[(size_t)w] > 0) staging[(size_t)w] = std::make_shared<dmrgx_host::DevBuffer>((size_t)stage_len[(size_t)w], dmrgx_host::DevBuffer::device_only_t{});
            size_t si = 0;
            std::vector<dmrgx_rdm_vec_task> vec_tasks;
            for (const auto& kv : per) {
                const PetscInt blk = kv.first, k = kv.second.first, cnt = kv.second.second, n = M[side]->Sizes(blk);
                rot->old_sector.push_back((int32_t)blk); rot->kept.push_back((int32_t)cnt);
                const int own = owner[(size_t)(2 * k + side)];
                const int64_t soff = stage_off[si++];
                if (own < 0) { rot->rt.push_back(nullptr); qn_list.push_back(M[side]->List(blk)); qn_size.push_back(cnt); continue; }   /* spectrum only */
                auto buf = std::make_shared<dmrgx_host::DevBuffer>((size_t)cnt * n, dmrgx_host::DevBuffer::device_only_t{});
                double* dst = W > 1 ? staging[(size_t)own]->dev_uninitialised() + soff : buf->dev_uninitialised();
                if (own == me) vec_tasks.push_back(dmrgx_rdm_vec_task{side, (int32_t)k, (int32_t)cnt, 0, dst, (int64_t)n});
                rot->rt.push_back(buf);
                qn_list.push_back(M[side]->List(blk)); qn_size.push_back(cnt);
            }
            /* the kept rows of all sectors of this side in one launch */
            if (!vec_tasks.empty() && dmrgx_rdm_eigenvectors_batch(rdm, (int32_t)vec_tasks.size(), vec_tasks.data(), nullptr)) { dmrgx_rdm_destroy(rdm); SETERRQ1(mpi_comm, 1, "dmrgx_rdm_eigenvectors_batch: %s", dmrgx_last_error()); }
            if (W > 1) {
                for (int w = 0; w < W; ++w)
                    if (staging[(size_t)w] && dmrgx_comm_bcast(comm, staging[(size_t)w]->dev_uninitialised(), (size_t)stage_len[(size_t)w] * sizeof(double), w, nullptr)) { dmrgx_rdm_destroy(rdm); SETERRQ1(mpi_comm, 1, "dmrgx_comm_bcast: %s", dmrgx_last_error()); }
                si = 0;
                size_t ri = 0;
                for (const auto& kv : per) {
                    const PetscInt blk = kv.first, k = kv.second.first, cnt = kv.second.second, n = M[side]->Sizes(blk);
                    const int own = owner[(size_t)(2 * k + side)];
                    const int64_t soff = stage_off[si++];
                    const auto& buf = rot->rt[ri++];
                    if (own < 0 || !buf) continue;
                    if (dmrgx_memcpy_d2d(buf->dev_uninitialised(), staging[(size_t)own]->dev_uninitialised() + soff, (size_t)cnt * (size_t)n * sizeof(double), nullptr)) { dmrgx_rdm_destroy(rdm); SETERRQ1(mpi_comm, 1, "dmrgx_memcpy_d2d: %s", dmrgx_last_error()); }
                }
                dmrgx_stream_sync(nullptr);                              /* the staging buffers go out of scope below */
            }
            BT[side]->RotMatT = std::make_shared<dmrgx_host::SectorMat>();
            BT[side]->RotMatT->rot = rot;
            ierr = BT[side]->QN.Initialize(mpi_comm, qn_list, qn_size); CHKERRQ(ierr);
        }
        /* The density-matrix object is destroyed one truncation LATER (one rank): its destruction reads the solver's verification of the kept
           eigenpairs (eigenvalues against the Rayleigh quotients of the finished vectors), which by then has long arrived -- destroying it
           here made the host wait for the GPU to drain (5 % of a configs[1] step, profiles/r05_hostprof_m512.txt) before it could queue the
           rotations.  Everything queued above is stream-ordered; the rotation rows live in buffers of their own. */
        if (W > 1) dmrgx_stream_sync(nullptr);
        dmrgx_rdm* old_rdm = W > 1 ? rdm : rdm_pending;
        rdm_pending = W > 1 ? nullptr : rdm;
        if (old_rdm && dmrgx_rdm_destroy(old_rdm)) SETERRQ1(mpi_comm, 1, "dmrgx_rdm_destroy: %s", dmrgx_last_error());
        return 0;
    }

    /* ---- operator residency (engine extension) ----------------------------------------------------------------------
       A site operator of a stored block can only be touched again by (a) an inter-block term of a later superblock --
       as a LEFT block (sites i with a term (i, j), j >= nb, in the lattice's term list; also covers the enlargement by
       site nb) or as a reflected RIGHT block (site s = nout-1-j of a term (i, j) cut between i and j in a superblock of
       nout sites) -- or (b) a registered correlator, measured on the two half-lattice blocks at the centre.  Everything
       else is neither rotated nor kept in HBM. */
    PetscErrorCode BuildNeedTables()
    {
        if (need_built) return 0;
        const PetscInt N = num_sites;
        need_left.assign((size_t)N + 1, std::vector<char>()); need_right.assign((size_t)N + 1, std::vector<char>());
        for (PetscInt nb = 1; nb <= N; ++nb) { need_left[(size_t)nb].assign((size_t)nb, 0); need_right[(size_t)nb].assign((size_t)nb, 0); }
        for (const Hamiltonians::Term& t : Ham.H(N)) {
            const PetscInt I = std::min(t.Isite, t.Jsite), J = std::max(t.Isite, t.Jsite);
            for (PetscInt nb = I + 1; nb <= J && nb <= N; ++nb) need_left[(size_t)nb][(size_t)I] = 1;
            /* right block of nb sites (+1 new site) in a superblock of nout = c + nb + 1 sites, cut c in (I, J): local site nb - (J - c) */
            for (PetscInt nb = 1; nb < N; ++nb)
                for (PetscInt c = I + 1; c < J; ++c) {
                    const PetscInt s = nb - (J - c);
                    if (s < 0 || s >= nb || c + nb + 1 > N) continue;
                    need_right[(size_t)nb][(size_t)s] = 1;
                }
        }
        /* On W ranks the correlators are dealt over the ranks (round 3, CorrelatorDealing.hpp): a rank measures only its share, so it
           carries only the site operators ITS correlators read on the way back to the centre (in round 2 every rank rotated all of
           them, up to 124 correlator-only operators per step at configs[3]); the values are summed over the ranks at the measurement. */
        {
            const int W = dmrgx_host::WorldComm() ? dmrgx_host::WorldSize() : 1;
            std::vector<std::vector<int64_t>> sites(measurements.size());
            for (size_t ic = 0; ic < measurements.size(); ++ic) {
                for (const Op& o : measurements[ic].SysOps) sites[ic].push_back((int64_t)o.idx);
                for (const Op& o : measurements[ic].EnvOps) sites[ic].push_back((int64_t)o.idx);
            }
            corr_owner = dmrgx_host::DealCorrelators(sites, (int64_t)N, W);      /* -1: measured by every rank (one-rank runs) */
        }
        const int me_rank = dmrgx_host::WorldComm() ? dmrgx_host::WorldRank() : 0;
        corr_sites.assign((size_t)N, 0);
        for (size_t ic = 0; ic < measurements.size(); ++ic) {
            if (corr_owner[ic] >= 0 && corr_owner[ic] != me_rank) continue;
            const Correlator& c = measurements[ic];
            for (const Op& o : c.SysOps) if (o.idx >= 0 && o.idx < N) corr_sites[(size_t)o.idx] = 1;
            for (const Op& o : c.EnvOps) if (o.idx >= 0 && o.idx < N) corr_sites[(size_t)o.idx] = 1;
        }
        need_built = true;
        return 0;
    }
    /** sites of a block with nb sites whose operators must stay resident */
    std::vector<char> NeedMask(PetscInt nb, bool with_corr, bool left_only)
    {
        BuildNeedTables();
        std::vector<char> m((size_t)nb, 0);
        if (nb < 1 || nb > num_sites) return std::vector<char>((size_t)std::max<PetscInt>(nb, 0), 1);
        for (PetscInt i = 0; i < nb; ++i) {
            m[(size_t)i] = need_left[(size_t)nb][(size_t)i] || (!left_only && need_right[(size_t)nb][(size_t)i]) || (with_corr && nb < num_sites / 2 && corr_sites[(size_t)i]);
        }
        /* never empty: an all-zero mask would read as "keep everything" */
        if (std::find(m.begin(), m.end(), 1) == m.end()) m[(size_t)nb - 1] = 1;
        return m;
    }
    /** an input block of a finished step is only read again through inter-block terms */
    PetscErrorCode PruneConsumed(PetscInt idx)
    {
        if (idx < 0 || idx >= (PetscInt)sys_blocks.size() || !sys_blocks[(size_t)idx].Initialized()) return 0;
        Block& b = sys_blocks[(size_t)idx];
        if (b.NumSites() < Ham.NumEnvSites() * 2) return 0;         /* the exact initial blocks stay whole */
        return b.PruneOperators(NeedMask(b.NumSites(), false, false));
    }

    /** One record per step in KronStats.json: the superblock as the plan sees it (SURVEY 8d quantities computed from the
        actual sector tables), the kept-sector tables of both input blocks, and -- with -step_profile -- the HIP-event time of
        the two GEMM stages summed over the step's MatMults. */
    PetscErrorCode SaveKronStats(const dmrgx_kron_info& ki, const Block& Sys, const Block& Env, PetscInt nterms, PetscInt nmatvec, double eigs_seconds,
                                 const double* ms4, int64_t napp)
    {
        if (mpi_rank || !fp_kron) return 0;
        fprintf(fp_kron, "%s  {\"GlobIdx\": %lld, \"LoopType\": \"%s\", \"NSites_Sys\": %lld, \"NSites_Env\": %lld, \"n_states\": %lld, \"flops_alg\": %.17g, \"bytes_alg\": %.17g, "
                         "\"flops_exec\": %.17g, \"bytes_workspace\": %.17g, \"n_groups\": %d, \"n_tiles_stage1\": %d, \"n_tiles_stage2\": %d, \"n_terms_all\": %lld, "
                         "\"matmults\": %lld, \"eigs_seconds\": %.9g, \"timed_applies\": %lld, \"ms_stage1\": %.9g, \"ms_stage2\": %.9g,\n",
                kron_rows ? ",\n" : "", LLD(GlobIdx), LoopType ? "Sweep" : "Warmup", LLD(Sys.NumSites()), LLD(Env.NumSites()), LLD(ki.n_states), ki.flops_alg, ki.bytes_alg, ki.flops_exec,
                ki.bytes_workspace, ki.n_groups, ki.n_tiles_stage1, ki.n_tiles_stage2, LLD(nterms), LLD(nmatvec), eigs_seconds, LLD(napp), ms4[0] + ms4[1], ms4[2] + ms4[3]);
        const Block* B[2] = {&Sys, &Env};
        const char* tag[2] = {"sys", "env"};
        for (int s = 0; s < 2; ++s) {
            const std::vector<PetscReal> q = B[s]->Magnetization.List();
            const std::vector<PetscInt> n = B[s]->Magnetization.Sizes();
            fprintf(fp_kron, "   \"%s_qn\": [", tag[s]);
            for (size_t i = 0; i < q.size(); ++i) fprintf(fp_kron, "%s%g", i ? ", " : "", q[i]);
            fprintf(fp_kron, "], \"%s_sizes\": [", tag[s]);
            for (size_t i = 0; i < n.size(); ++i) fprintf(fp_kron, "%s%lld", i ? ", " : "", LLD(n[i]));
            fprintf(fp_kron, "]%s", s == 0 ? ",\n" : "}");
        }
        fflush(fp_kron);
        ++kron_rows;
        return 0;
    }

private:
    struct StepData {
        PetscInt NumSites_Sys = 0, NumSites_Env = 0, NumSites_SysEnl = 0, NumSites_EnvEnl = 0;
        PetscInt NumStates_Sys = 0, NumStates_Env = 0, NumStates_SysEnl = 0, NumStates_EnvEnl = 0, NumStates_SysRot = 0, NumStates_EnvRot = 0, NumStates_H = 0;
        PetscScalar GSEnergy = 0; PetscReal TruncErr_Sys = 0, TruncErr_Env = 0;
    };
    struct TimingsData { PetscLogDouble tEnlr = 0, tKron = 0, tDiag = 0, tRdms = 0, tRotb = 0, Total = 0; PetscInt nMatMult = 0, nRotOps = 0;
                         double msAllGather = 0, msApply = 0; };      /* (ranks > 1) HIP-event time of the step's all-gathers / applies on this rank */
    typedef enum { WarmupStep = 0, SweepStep = 1, NullStep = -1 } Step_t;
    typedef enum { SWEEP_MODE_NULL, SWEEP_MODE_NSWEEPS, SWEEP_MODE_MSWEEPS, SWEEP_MODE_TOLERANCE_TEST } SweepMode_t;

    Block& AddSite() { return SingleSite; }

    void PrintBlocks(const PetscInt& nsys, const PetscInt& nenv) const { printf("  [%lld]-* *-[%lld]\n", LLD(nsys), LLD(nenv)); }

    PetscErrorCode SaveStepHeaders()
    {
        fprintf(fp_step, "{\n  \"headers\" : [\"GlobIdx\", \"LoopType\", \"LoopIdx\", \"StepIdx\", \"NSites_Sys\", \"NSites_Env\", \"NSites_SysEnl\", \"NSites_EnvEnl\", "
                         "\"NStates_Sys\", \"NStates_Env\", \"NStates_SysEnl\", \"NStates_EnvEnl\", \"NStates_SysRot\", \"NStates_EnvRot\", \"NumStates_H\", "
                         "\"TruncErr_Sys\", \"TruncErr_Env\", \"GSEnergy\"  ],\n  \"table\" : ");
        return 0;
    }
    PetscErrorCode SaveStepData(const StepData& d)
    {
        fprintf(fp_step, "%s    [ %lld, %s, %lld, %lld, %lld, %lld, %lld, %lld, %lld, %lld, %lld, %lld, %lld, %lld, %lld, %.12g, %.12g, %.12g]", rows_written ? ",\n" : "",
                LLD(GlobIdx), LoopType ? "\"Sweep\"" : "\"Warmup\"", LLD(LoopIdx), LLD(StepIdx), LLD(d.NumSites_Sys), LLD(d.NumSites_Env), LLD(d.NumSites_SysEnl), LLD(d.NumSites_EnvEnl),
                LLD(d.NumStates_Sys), LLD(d.NumStates_Env), LLD(d.NumStates_SysEnl), LLD(d.NumStates_EnvEnl), LLD(d.NumStates_SysRot), LLD(d.NumStates_EnvRot), LLD(d.NumStates_H),
                d.TruncErr_Sys, d.TruncErr_Env, d.GSEnergy);
        fflush(fp_step);
        return 0;
    }
    PetscErrorCode SaveTimingsHeaders()
    {
        fprintf(fp_timings, "{\n  \"headers\" : [\"GlobIdx\", \"Total\", \"Enlr\", \"Kron\", \"Diag\", \"Rdms\", \"Rotb\", \"MatMults\", \"RotOps\", \"AllGatherMs\", \"ApplyMs\" ],\n  \"table\" : ");
        return 0;
    }
    PetscErrorCode SaveTimingsData(const TimingsData& d)
    {
        fprintf(fp_timings, "%s    [ %lld, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %lld, %lld, %.6g, %.6g ]", rows_written ? ",\n" : "", LLD(GlobIdx), d.Total, d.tEnlr, d.tKron, d.tDiag, d.tRdms, d.tRotb, LLD(d.nMatMult), LLD(d.nRotOps), d.msAllGather, d.msApply);
        fflush(fp_timings);
        return 0;
    }
    /** Per-sector RDM eigenvalues of one side, in the reference's EntanglementSpectra.json layout. */
    PetscErrorCode SaveEntanglementSpectrum(int side, const std::vector<Eigen_t>& eigen, const QuantumNumbers& qn)
    {
        if (!fp_entanglement || mpi_rank) return 0;
        /* what the record needs, by value: (sector quantum number, eigenvalue) in the order given */
        auto rec = std::make_shared<std::vector<std::pair<double, double>>>();
        rec->reserve(eigen.size());
        for (const Eigen_t& e : eigen) rec->push_back({(double)qn.List(e.blkIdx), (double)e.eigval});
        auto blk = std::make_shared<std::vector<PetscInt>>();
        blk->reserve(eigen.size());
        for (const Eigen_t& e : eigen) blk->push_back(e.blkIdx);
        FILE* fp = fp_entanglement;
        const bool first_row = rows_written == 0;
        const long long glob = (long long)GlobIdx;
        spectra_writer.Push([fp, side, first_row, glob, rec, blk] {
            if (side == 0) fprintf(fp, "%s  {\n    \"GlobIdx\": %lld,\n", first_row ? "" : ",\n", glob);
            fprintf(fp, "    \"%s\": [\n", side == 0 ? "Sys" : "Env");
            PetscInt prev = -1; bool first_sector = true;
            for (size_t i = 0; i < rec->size(); ++i) {
                if ((*blk)[i] != prev) {
                    if (!first_sector) fprintf(fp, "] },\n");
                    fprintf(fp, "      {\"sector\": %g, \"vals\": [ %g", (*rec)[i].first, (*rec)[i].second);
                    prev = (*blk)[i]; first_sector = false;
                } else fprintf(fp, ", %g", (*rec)[i].second);
            }
            if (!first_sector) fprintf(fp, "] }\n");
            fprintf(fp, side == 0 ? "    ],\n" : "    ]\n  }");
            fflush(fp);
        });
        return 0;
    }

    MPI_Comm mpi_comm;
    PetscMPIInt mpi_rank = 0, mpi_size = 1;
    PetscBool debug_symm = PETSC_FALSE;
    PetscBool init = PETSC_FALSE, verbose = PETSC_FALSE, no_symm = PETSC_FALSE, do_shell = PETSC_TRUE, dry_run = PETSC_FALSE, warmed_up = PETSC_FALSE;
    PetscReal qn_sector = 0.0;
    PetscReal eps_tol = 1.0e-8;     /* SLEPc's default relative residual tolerance */
    PetscInt eps_ncv = 0, eps_max_it = 1000, eps_gd_minv = 0;      /* 0: the library's default for the solver type (ncv: 16 krylovschur, 8 gd; minv 1) */
    int32_t eps_method = 0;         /* -H_eps_type: 0 krylovschur (thick-restart Lanczos), 1 gd (generalized Davidson, diagonal preconditioner) */
    std::string scratch_dir, data_dir;
    FILE *fp_step = NULL, *fp_timings = NULL, *fp_entanglement = NULL, *fp_data = NULL, *fp_corr = NULL, *fp_kron = NULL;
    BackgroundWriter spectra_writer;           /* EntanglementSpectra.json records are formatted and written off the critical path */
    PetscInt kron_rows = 0;
    PetscBool prune_ops = PETSC_TRUE;          /* -prune_ops 0: rotate and keep every site operator of every block, as the reference does */
    PetscBool step_profile = PETSC_FALSE;      /* -step_profile 1: HIP-event timing of the GEMM stages of every MatMult (KronStats.json) */
    PetscInt rot_ops_total = 0;         /**< site operators this rank rotated so far */
    bool need_built = false;
    std::vector<std::vector<char>> need_left, need_right;
    std::vector<char> corr_sites;
    PetscInt mwarmup = 0, nsweeps = 0, msweep_idx = 0, opt_min_block = PETSC_DEFAULT;
    std::vector<PetscInt> msweeps, maxnsweeps, sweeps_mstates;
    SweepMode_t sweep_mode = SWEEP_MODE_NULL;
    Hamiltonian Ham;
    Block SingleSite;
    PetscInt num_sites = 0, num_sys_blocks = 0, sys_ninit = 0;
    std::vector<Block> sys_blocks;
    Step_t LoopType = NullStep;
    PetscInt GlobIdx = 0, LoopIdx = 0, StepIdx = 0;
    PetscScalar gse = 0.0;
    std::vector<PetscReal> trunc_err;
    PetscLogDouble t0abs = 0.0;
    PetscInt total_matmults = 0, last_sweep_steps = 0, last_sweep_matmults = 0;
    double total_eigs_seconds = 0.0, last_sweep_seconds = 0.0;
    size_t device_bytes_after_sweep = 0;
    struct Correlator {
        PetscInt idx = 0;
        std::vector<Op> SysOps, EnvOps;     /**< operators on the (enlarged) system / environment block, block-local site index */
        std::string name, desc1, desc2, desc3;
    };
    std::vector<Correlator> measurements;
    std::vector<int> corr_owner;        /**< per correlator: the rank that measures it, -1 = every rank (BuildNeedTables) */
    PetscBool corr_headers_printed = PETSC_FALSE, corr_printed_first = PETSC_FALSE;
    PetscBool do_scratch_dir = PETSC_FALSE, restart = PETSC_FALSE, restart_options = PETSC_FALSE;
    std::string restart_dir;
    PetscInt restart_sys_ninit = 0, restart_num_sites = 0, restart_msweep_idx = -1;
    PetscInt rows_written = 0;          /**< steps recorded by THIS run (a restarted run starts at GlobIdx > 0) */

    /* ---- start vector of the eigensolve from the previous step's ground state (White's wavefunction transformation) ---- */
    struct PrevStep {
        bool valid = false;
        PetscInt insys = -1, inenv = -1, outsys = -1, outenv = -1;
        Vec psi;
        std::vector<std::array<PetscInt, 5>> kb;                   /**< (IL, IR, nL, nR, offset) of every KronBlock */
        std::vector<std::vector<EnlPart>> partsL, partsR;            /**< decomposition of the enlarged sectors */
        std::shared_ptr<dmrgx_host::BasisRotation> rotL, rotR;     /**< this step's truncations */
    } prev;
    std::vector<std::shared_ptr<dmrgx_host::BasisRotation>> block_rot;   /**< rotation that created sys_blocks[i] from sys_blocks[i-1] (x) site */
    /* Which VERSION of sys_blocks[i-1] that was.  The warm-up re-derives the environment blocks in decreasing order, so a stored
       block i is in general the child of a version of block i-1 that has been overwritten since; the overlap between the two
       versions' bases is carried along (RecordBlockBasis) so that the stored rotation can still be read in the current basis. */
    std::vector<RotMeta> rot_meta;
    std::vector<int64_t> block_ver;                       /**< current version of every stored block (0: as initialised) */
    std::vector<std::shared_ptr<BasisOverlap>> block_ovl; /**< [i]: parent version of block_rot[i+1] -> current version of block i */
    int64_t ver_counter = 0;
    dmrgx_rdm* rdm_pending = nullptr;          /**< the previous truncation's density matrices: destroyed (and their verification read) one truncation later */
    /** which path the density-matrix solver took, per truncation (dmrgx_rdm_info): DMRGRun.json RdmCalls ... RdmMaxWyBlocks */
    PetscInt rdm_calls = 0, rdm_jacobi_calls = 0, trid_persistent_calls = 0, trid_launch_calls = 0, trid_fallbacks = 0, trid_max_wgs = 0, rdm_max_levels = 0, rdm_max_wy_blocks = 0;
    PetscInt guesses_projected = 0, guesses_rejected = 0;                       /**< start vectors that went through a basis overlap */
    PetscBool use_guess = PETSC_TRUE;
    PetscBool use_guess_overlap = PETSC_TRUE;   /* -wavefunction_guess_overlap 0: no start vector where the stored chain of bases is broken (round-2 behaviour) */
    PetscInt guesses_used = 0;
    /* eigenbases of the density matrices at the previous visit of every block: warm start of the Jacobi eigensolver */
    struct WarmBasis { std::vector<int32_t> sizes; std::map<int32_t, std::shared_ptr<dmrgx_host::DevBuffer>> E; };
    std::map<std::pair<PetscInt, int>, WarmBasis> rdm_basis;
    PetscBool use_rdm_warm = PETSC_FALSE;
    PetscBool use_corr_batch = PETSC_TRUE;      /* -corr_batch 0: every correlator through its own MatMult + dot, as the reference does */
};

#endif
