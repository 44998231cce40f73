/** @file DMRGBlock.hpp
    Block::SpinBase -- a block of spin sites with its per-site Sz(i), Sp(i) (Sm(i) as a transposed view), its block
    Hamiltonian H and its Magnetization sectors.  Same public interface, conventions and error codes as the reference
    class (reference include/DMRGBlock.hpp:79-434, src/DMRGBlock.cpp), re-implemented on device-resident sector cells:
      - operators are SectorMat handles (HBM), copies of a block are shallow and Destroy() on one copy invalidates the
        other (reference tests/UnitTests_DMRGBlock.cpp:56-70);
      - RotateOperators is ONE call of dmrgx_rotate_ops for all 2*nsites+1 operators (reference loops MatMatMatMult,
        src/DMRGBlock.cpp:763-772);
      - the disk-spill entry points exist with the reference's names but keep the block resident (288 GB of HBM3E):
        scratch/restart files are outside the hot path. */
#ifndef DMRGX_DMRGBLOCK_HPP
#define DMRGX_DMRGBLOCK_HPP

#include <map>
#include <string>
#include <vector>
#include <stdexcept>
#include "petsc_compat.hpp"
#include "SectorMat.hpp"
#include "QuantumNumbers.hpp"

/** Operator type == the shift it applies to the (descending) sector index of the column block */
typedef enum { OpSm = -1, OpSz = 0, OpSp = +1, OpEye = +2 } Op_t;
static const std::map<Op_t, std::string> OpString = {{OpSm, "Sm"}, {OpSz, "Sz"}, {OpSp, "Sp"}};
#define OpToCStr(OP) ((OpString.find(OP)->second).c_str())
#define OpToStr(OP) (OpString.find(OP)->second)
static const std::vector<Op_t> BasicOpTypes = {OpSz, OpSp};

typedef enum { SideLeft = 0, SideRight = 1 } Side_t;
static const std::vector<Side_t> SideTypes = {SideLeft, SideRight};

typedef enum { SpinOneHalf = 102, SpinOne = 101, SpinNull = -2 } Spin_t;
static const std::map<std::string, Spin_t> SpinTypes = {{"1/2", SpinOneHalf}, {"1", SpinOne}};

namespace Block {

class SpinBase
{
public:
    /** Sectors of the block basis (named as in the reference) */
    QuantumNumbers Magnetization;
    /** Block Hamiltonian (operators acting inside the block) */
    Mat H = nullptr;

    virtual ~SpinBase() {}
    virtual PetscInt loc_dim() const { return _loc_dim; }
    virtual std::vector<PetscScalar> loc_qn_list() const { return _loc_qn_list; }
    virtual std::vector<PetscInt> loc_qn_size() const { return _loc_qn_size; }

    PetscBool MPIInitialized() const { return mpi_init; }
    PetscBool Initialized() const { return init; }
    PetscBool Saved() const { return saved; }
    MPI_Comm MPIComm() const { return mpi_comm; }
    PetscInt NumSites() const { return num_sites; }
    PetscInt NumStates() const { return num_states; }

    PetscErrorCode Initialize(const MPI_Comm& comm_in)
    {
        if (mpi_init) SETERRQ(mpi_comm, 1, "This initializer should only be called once.");
        mpi_comm = comm_in;
        MPI_Comm_rank(mpi_comm, &mpi_rank); MPI_Comm_size(mpi_comm, &mpi_size);
        mpi_init = PETSC_TRUE;
        return 0;
    }

    /** num_states_in == PETSC_DEFAULT: exact block of num_sites_in sites; one site gets the spin operators. */
    PetscErrorCode Initialize(const MPI_Comm& comm_in, const PetscInt& num_sites_in, const PetscInt& num_states_in, const PetscBool& init_ops = PETSC_TRUE)
    {
        PetscErrorCode ierr;
        ierr = PetscOptionsGetBool(NULL, NULL, "-verbose", &verbose, NULL); CHKERRQ(ierr);
        if (!mpi_init) { ierr = Initialize(comm_in); CHKERRQ(ierr); }
        else if (comm_in != mpi_comm) SETERRQ(PETSC_COMM_SELF, 1, "Mismatch in MPI communicators.");
        {
            char spin[10]; PetscBool set;
            ierr = PetscOptionsGetString(NULL, NULL, "-spin", spin, 10, &set); CHKERRQ(ierr);
            if (set) {
                auto it = SpinTypes.find(std::string(spin));
                if (it == SpinTypes.end()) SETERRQ1(mpi_comm, 1, "Given -spin %s not valid/implemented.", spin);
                spin_type = it->second;
                if (spin_type == SpinOne) { _loc_dim = 3; _loc_qn_list = {+1.0, 0.0, -1.0}; _loc_qn_size = {1, 1, 1}; }
                else { _loc_dim = 2; _loc_qn_list = {+0.5, -0.5}; _loc_qn_size = {1, 1}; }
            }
        }
        num_sites = num_sites_in;
        if (num_states_in == PETSC_DEFAULT) { num_states = 1; for (PetscInt i = 0; i < num_sites; ++i) num_states *= loc_dim(); }
        else num_states = num_states_in;
        SzData.assign((size_t)num_sites, nullptr); SpData.assign((size_t)num_sites, nullptr); SmData.assign((size_t)num_sites, nullptr);
        init = PETSC_TRUE; init_once = PETSC_TRUE; init_Sm = PETSC_FALSE; ops_pruned = PETSC_FALSE; dead = PETSC_FALSE;
        if (!init_ops && num_sites > 0) {}
        else if (init_ops && num_sites == 1) {
            ierr = Magnetization.Initialize(mpi_comm, loc_qn_list(), loc_qn_size()); CHKERRQ(ierr);
            ierr = MatSpinSzCreate(SzData[0]); CHKERRQ(ierr);
            ierr = MatSpinSpCreate(SpData[0]); CHKERRQ(ierr);
            H = dmrgx_host::SectorMat::Dense(0, Magnetization.Sizes32());    /* single-site Hamiltonian: zero */
            ierr = CheckSectors(); CHKERRQ(ierr);
        }
        else if (init_ops && num_sites > 1) { /* operators are created once the sectors are known (see below) */ }
        else SETERRQ1(mpi_comm, PETSC_ERR_ARG_OUTOFRANGE, "Invalid input num_sites_in > 0. Given %lld.", LLD(num_sites_in));
        return 0;
    }

    PetscErrorCode Initialize(const MPI_Comm& comm_in, const PetscInt& num_sites_in, const std::vector<PetscReal>& qn_list_in,
                              const std::vector<PetscInt>& qn_size_in, const PetscBool& init_ops = PETSC_TRUE)
    {
        QuantumNumbers tmp;
        PetscErrorCode ierr = tmp.Initialize(comm_in, qn_list_in, qn_size_in); CHKERRQ(ierr);
        ierr = Initialize(comm_in, num_sites_in, tmp.NumStates(), init_ops); CHKERRQ(ierr);
        Magnetization = tmp;
        if (init_ops && num_sites_in > 1) {
            /* empty (zero) operators with one dense cell per admissible sector block, ready for MatSetValues-style fills */
            for (PetscInt i = 0; i < num_sites; ++i) {
                SzData[i] = dmrgx_host::SectorMat::Dense(OpSz, Magnetization.Sizes32());
                SpData[i] = dmrgx_host::SectorMat::Dense(OpSp, Magnetization.Sizes32());
            }
        }
        return 0;
    }

    PetscErrorCode Initialize(const PetscInt& num_sites_in, const QuantumNumbers& qn_in)
    {
        PetscErrorCode ierr = qn_in.CheckInitialized(); CHKERRQ(ierr);
        ierr = Initialize(qn_in.MPIComm(), num_sites_in, qn_in.NumStates(), PETSC_FALSE); CHKERRQ(ierr);
        Magnetization = qn_in;
        return 0;
    }

    /* ---- checkpoint files ----------------------------------------------------------------
       Same directory layout and text files as the reference (src/DMRGBlock.cpp:889-971): BlockInfo.dat,
       QuantumNumbers.dat, Sz_%09d.mat, Sp_%09d.mat, H_%09d.mat.  The operator files hold this engine's sector-cell
       form (little-endian): int64 {magic "DMRGXOP1", shift, nsec, sizes[nsec], ncells}, then per cell int64 {row
       sector, r0, c0, nr, nc, kind}, double scale and, for dense cells, nr*nc doubles row-major.  (The reference writes
       PETSc's binary MatView format, which is not part of its source tree.) */
    static std::string OpFilename(const std::string& dir, const std::string& name, const size_t isite = 0)
    {
        char buf[64];
        snprintf(buf, sizeof(buf), "%s_%09zu.mat", name.c_str(), isite);
        return dir + buf;
    }
    static PetscErrorCode WriteOperatorFile(const std::string& fn, const Mat& op)
    {
        if (!op || op->transpose_of || op->plan) return PETSC_ERR_ARG_WRONG;
        FILE* fp = fopen(fn.c_str(), "wb");
        if (!fp) return PETSC_ERR_FILE_OPEN;
        auto put = [&](int64_t v) { fwrite(&v, sizeof(v), 1, fp); };
        put(0x31504f5847524d44ll); put(op->shift); put((int64_t)op->sizes.size());
        for (int32_t sz : op->sizes) put(sz);
        put((int64_t)op->cells.size());
        for (dmrgx_host::MatCell& c : op->cells) {
            put(c.q); put(c.r0); put(c.c0); put(c.nr); put(c.nc); put(c.kind);
            fwrite(&c.scale, sizeof(double), 1, fp);
            if (c.kind == DMRGX_CELL_DENSE) {
                const double* h = c.buf->host_ro() + c.off;              /* D2H of the owning buffer on first touch */
                for (int32_t i = 0; i < c.nr; ++i) fwrite(h + (int64_t)i * c.ld, sizeof(double), (size_t)c.nc, fp);
            }
        }
        const bool ok = !ferror(fp);
        fclose(fp);
        return ok ? 0 : PETSC_ERR_FILE_OPEN;
    }
    static PetscErrorCode ReadOperatorFile(const std::string& fn, Mat& op)
    {
        FILE* fp = fopen(fn.c_str(), "rb");
        if (!fp) return PETSC_ERR_FILE_OPEN;
        bool ok = true;
        auto get = [&]() { int64_t v = 0; if (fread(&v, sizeof(v), 1, fp) != 1) ok = false; return v; };
        op = std::make_shared<dmrgx_host::SectorMat>();
        if (get() != 0x31504f5847524d44ll) { fclose(fp); return PETSC_ERR_FILE_OPEN; }
        op->shift = (int32_t)get();
        const int64_t ns = get();
        for (int64_t i = 0; ok && i < ns; ++i) op->sizes.push_back((int32_t)get());
        const int64_t nc = get();
        for (int64_t i = 0; ok && i < nc; ++i) {
            dmrgx_host::MatCell c;
            c.q = (int32_t)get(); c.r0 = (int32_t)get(); c.c0 = (int32_t)get(); c.nr = (int32_t)get(); c.nc = (int32_t)get(); c.kind = (int32_t)get();
            if (fread(&c.scale, sizeof(double), 1, fp) != 1) ok = false;
            if (ok && c.kind == DMRGX_CELL_DENSE) {
                if (c.nr < 0 || c.nc < 0) { ok = false; break; }
                c.ld = c.nc; c.off = 0;
                c.buf = std::make_shared<dmrgx_host::DevBuffer>((size_t)c.nr * c.nc);
                if (fread(c.buf->host(), sizeof(double), (size_t)c.nr * c.nc, fp) != (size_t)c.nr * c.nc) ok = false;
            }
            op->cells.push_back(c);
        }
        fclose(fp);
        return ok ? 0 : PETSC_ERR_FILE_OPEN;
    }

    /** Writes the block into `dir` (which must exist) without destroying it: the checkpoint copy of SaveAndDestroy. */
    PetscErrorCode SaveToDisk(const std::string& dir_in)
    {
        if (!init) SETERRQ(mpi_comm, PETSC_ERR_ARG_CORRUPT, "Block not yet initialized.");
        std::string dir = dir_in;
        if (dir.empty()) SETERRQ(mpi_comm, 1, "Save dir cannot be empty.");
        if (dir.back() != '/') dir += '/';
        PetscBool flg = PETSC_FALSE;
        PetscErrorCode ierr = PetscTestDirectory(dir.c_str(), 'r', &flg); CHKERRQ(ierr);
        if (!flg) SETERRQ1(mpi_comm, 1, "Directory %s does not exist. Please verify that -scratch_dir is specified correctly.", dir.c_str());
        /* pruned site operators (PruneOperators) have no file; InitializeFromDisk restores them as absent */
        for (PetscInt i = 0; i < num_sites; ++i) { if (!SzData[(size_t)i] && ops_pruned) continue; if (WriteOperatorFile(OpFilename(dir, "Sz", (size_t)i), SzData[(size_t)i])) SETERRQ1(mpi_comm, PETSC_ERR_FILE_OPEN, "cannot write %s", OpFilename(dir, "Sz", (size_t)i).c_str()); }
        for (PetscInt i = 0; i < num_sites; ++i) { if (!SpData[(size_t)i] && ops_pruned) continue; if (WriteOperatorFile(OpFilename(dir, "Sp", (size_t)i), SpData[(size_t)i])) SETERRQ1(mpi_comm, PETSC_ERR_FILE_OPEN, "cannot write %s", OpFilename(dir, "Sp", (size_t)i).c_str()); }
        if (H) { if (WriteOperatorFile(OpFilename(dir, "H", 0), H)) SETERRQ1(mpi_comm, PETSC_ERR_FILE_OPEN, "cannot write %s", OpFilename(dir, "H", 0).c_str()); }
        {
            FILE* fp = fopen((dir + "BlockInfo.dat").c_str(), "w");
            if (!fp) SETERRQ1(mpi_comm, PETSC_ERR_FILE_OPEN, "cannot write %sBlockInfo.dat", dir.c_str());
            fprintf(fp, "%-30s %zu\n%-30s %zu\n%-30s %d\n%-30s %d\n%-30s %lld\n%-30s %lld\n%-30s %lld\n", "NumBytesPetscInt", sizeof(PetscInt),
                    "NumBytesPetscScalar", sizeof(PetscScalar), "PetscUseComplex", 0, "SpinTypeKey", (int)spin_type, "NumSites", LLD(num_sites),
                    "NumStates", LLD(num_states), "NumSectors", LLD(Magnetization.NumSectors()));
            if (ops_pruned) fprintf(fp, "%-30s %d\n", "OpsPruned", 1);
            fclose(fp);
        }
        {
            FILE* fp = fopen((dir + "QuantumNumbers.dat").c_str(), "w");
            if (!fp) SETERRQ1(mpi_comm, PETSC_ERR_FILE_OPEN, "cannot write %sQuantumNumbers.dat", dir.c_str());
            const std::vector<PetscInt> sz = Magnetization.Sizes();
            const std::vector<PetscReal> ql = Magnetization.List();
            for (size_t i = 0; i < sz.size(); ++i) fprintf(fp, "%lld %.17g\n", LLD(sz[i]), ql[i]);
            fclose(fp);
        }
        return 0;
    }

    /** Rebuilds a block from the files written by SaveToDisk (InitializeFromDisk of the reference,
        src/DMRGBlock.cpp:184-300: BlockInfo.dat and QuantumNumbers.dat are cross-checked). */
    PetscErrorCode InitializeFromDisk(const MPI_Comm& comm_in, const std::string& dir_in)
    {
        std::string dir = dir_in;
        if (dir.empty()) SETERRQ(comm_in, 1, "Block directory cannot be empty.");
        if (dir.back() != '/') dir += '/';
        std::map<std::string, long long> info;
        {
            FILE* fp = fopen((dir + "BlockInfo.dat").c_str(), "r");
            if (!fp) SETERRQ1(comm_in, PETSC_ERR_FILE_OPEN, "cannot read %sBlockInfo.dat", dir.c_str());
            char key[128]; long long val;
            while (fscanf(fp, "%127s %lld", key, &val) == 2) info[key] = val;
            fclose(fp);
        }
        for (const char* k : {"NumBytesPetscInt", "NumBytesPetscScalar", "PetscUseComplex", "NumSites", "NumStates", "NumSectors"})
            if (!info.count(k)) SETERRQ2(comm_in, 1, "%sBlockInfo.dat: key %s missing.", dir.c_str(), k);
        if (info["NumBytesPetscInt"] != (long long)sizeof(PetscInt) || info["NumBytesPetscScalar"] != (long long)sizeof(PetscScalar) || info["PetscUseComplex"] != 0)
            SETERRQ1(comm_in, 1, "%sBlockInfo.dat was written with incompatible scalar/integer types.", dir.c_str());
        std::vector<PetscReal> ql; std::vector<PetscInt> qs;
        {
            FILE* fp = fopen((dir + "QuantumNumbers.dat").c_str(), "r");
            if (!fp) SETERRQ1(comm_in, PETSC_ERR_FILE_OPEN, "cannot read %sQuantumNumbers.dat", dir.c_str());
            long long sz; double q;
            while (fscanf(fp, "%lld %lf", &sz, &q) == 2) { qs.push_back((PetscInt)sz); ql.push_back(q); }
            fclose(fp);
        }
        if ((long long)ql.size() != info["NumSectors"]) SETERRQ2(comm_in, 1, "QuantumNumbers.dat has %lld sectors, BlockInfo.dat says %lld.", LLD(ql.size()), info["NumSectors"]);
        PetscErrorCode ierr = Initialize(comm_in, (PetscInt)info["NumSites"], ql, qs, PETSC_FALSE); CHKERRQ(ierr);
        if (num_states != (PetscInt)info["NumStates"]) SETERRQ2(comm_in, 1, "Sector sizes add up to %lld states, BlockInfo.dat says %lld.", LLD(num_states), info["NumStates"]);
        const bool pruned_on_disk = info.count("OpsPruned") && info["OpsPruned"] != 0;
        for (PetscInt i = 0; i < num_sites; ++i) {
            if (pruned_on_disk) {       /* a checkpoint of a pruned block holds the resident site operators only */
                PetscBool hz = PETSC_FALSE, hp = PETSC_FALSE;
                ierr = PetscTestFile(OpFilename(dir, "Sz", (size_t)i).c_str(), 'r', &hz); CHKERRQ(ierr);
                ierr = PetscTestFile(OpFilename(dir, "Sp", (size_t)i).c_str(), 'r', &hp); CHKERRQ(ierr);
                if (!hz || !hp) { ops_pruned = PETSC_TRUE; continue; }
            }
            if (ReadOperatorFile(OpFilename(dir, "Sz", (size_t)i), SzData[(size_t)i])) SETERRQ1(comm_in, PETSC_ERR_FILE_OPEN, "cannot read %s", OpFilename(dir, "Sz", (size_t)i).c_str());
            if (ReadOperatorFile(OpFilename(dir, "Sp", (size_t)i), SpData[(size_t)i])) SETERRQ1(comm_in, PETSC_ERR_FILE_OPEN, "cannot read %s", OpFilename(dir, "Sp", (size_t)i).c_str());
        }
        if (pruned_on_disk) ops_pruned = PETSC_TRUE;
        {   /* a block without a Hamiltonian (fixtures) has no H file */
            PetscBool has_h = PETSC_FALSE;
            ierr = PetscTestFile(OpFilename(dir, "H", 0).c_str(), 'r', &has_h); CHKERRQ(ierr);
            if (has_h && ReadOperatorFile(OpFilename(dir, "H", 0), H)) SETERRQ1(comm_in, PETSC_ERR_FILE_OPEN, "cannot read %s", OpFilename(dir, "H", 0).c_str());
        }
        ierr = CheckOperatorBlocks(); CHKERRQ(ierr);
        return 0;
    }

    /* ---- scratch storage: names kept, blocks stay in HBM ------------------------------------------------ */
    PetscErrorCode InitializeSave(const std::string& save_dir_in) { save_dir = save_dir_in; init_save = PETSC_TRUE; return 0; }
    PetscErrorCode SetDiskStorage(const std::string& read_dir_in, const std::string& write_dir_in) { read_dir = read_dir_in; write_dir = write_dir_in; save_dir = read_dir_in; disk_set = PETSC_TRUE; return 0; }
    std::string SaveDir() const { return save_dir; }
    PetscBool SaveInitialized() const { return init_save; }
    PetscErrorCode SaveAndDestroy() { saved = PETSC_TRUE; return 0; }
    PetscErrorCode Retrieve() { saved = PETSC_FALSE; return 0; }
    PetscErrorCode EnsureSaved() { return 0; }
    PetscErrorCode EnsureRetrieved() { return 0; }

    /** Releases every operator; shallow copies of this block see emptied handles. */
    PetscErrorCode Destroy()
    {
        if (PetscUnlikely(!init)) return 0;
        for (PetscInt i = 0; i < num_sites; ++i) { MatDestroy(&SzData[i]); MatDestroy(&SpData[i]); }
        MatDestroy(&H);
        if (init_Sm) DestroySm();
        init = PETSC_FALSE;
        return 0;
    }

    /* ---- accessors ------------------------------------------------------------------------------------------ */
    Mat Sz(const PetscInt& Isite) const { if (Isite >= num_sites) throw std::runtime_error("Attempted to access non-existent site."); return SzData[Isite]; }
    Mat Sp(const PetscInt& Isite) const { if (Isite >= num_sites) throw std::runtime_error("Attempted to access non-existent site."); return SpData[Isite]; }
    Mat Sm(const PetscInt& Isite) const
    {
        if (Isite >= num_sites) throw std::runtime_error("Attempted to access non-existent site.");
        if (!init_Sm) throw std::runtime_error("Sm matrices were not initialized. Call CreateSm() first.");
        return SmData[Isite];
    }
    const std::vector<Mat>& Sz() const { return SzData; }
    const std::vector<Mat>& Sp() const { return SpData; }
    const std::vector<Mat>& Sm() const { return SmData; }
    /** engine-internal: replace an operator handle (enlargement / rotation) */
    void SetOp(Op_t op, PetscInt isite, const Mat& m) { (op == OpSz ? SzData : SpData)[(size_t)isite] = m; }

    /* ---- validity checks (same codes as the reference) -------------------------------------------------------- */
    PetscErrorCode CheckOperatorArray(const Op_t& OpType) const
    {
        const std::vector<Mat>* Op;
        switch (OpType) { case OpSm: Op = &SmData; break; case OpSz: Op = &SzData; break; case OpSp: Op = &SpData; break;
            default: SETERRQ(mpi_comm, PETSC_ERR_ARG_WRONG, "Incorrect operator type."); }
        for (PetscInt i = 0; i < num_sites; ++i) {
            if (!(*Op)[i] && ops_pruned) continue;       /* not resident: pruned by the sweep schedule (see PruneOperators) */
            if (!(*Op)[i]) SETERRQ2(mpi_comm, PETSC_ERR_ARG_CORRUPT, "%s[%lld] matrix not yet created.", OpToCStr(OpType), LLD(i));
            if ((*Op)[i]->N() != num_states)
                SETERRQ4(mpi_comm, PETSC_ERR_ARG_WRONG, "%s[%lld] matrix dimension does not match the number of states. Expected %lld. Got %lld.",
                         OpToCStr(OpType), LLD(i), LLD(num_states), LLD((*Op)[i]->N()));
        }
        return 0;
    }
    PetscErrorCode CheckOperators() const
    {
        if (!init) SETERRQ(mpi_comm, PETSC_ERR_ARG_CORRUPT, "Block not yet initialized.");
        if (dead) SETERRQ(mpi_comm, PETSC_ERR_ARG_WRONGSTATE, "The operators of this block were never computed: the sweep schedule marked it as a block that no later step reads.");
        PetscErrorCode ierr = CheckOperatorArray(OpSz); CHKERRQ(ierr);
        ierr = CheckOperatorArray(OpSp); CHKERRQ(ierr);
        if (init_Sm) { ierr = CheckOperatorArray(OpSm); CHKERRQ(ierr); }
        return 0;
    }
    PetscErrorCode CheckSectors() const
    {
        if (!init_once) SETERRQ(mpi_comm, PETSC_ERR_ARG_CORRUPT, "Block not yet initialized.");
        PetscErrorCode ierr = Magnetization.CheckInitialized(); CHKERRQ(ierr);
        if (num_states != Magnetization.NumStates())
            SETERRQ2(mpi_comm, PETSC_ERR_ARG_WRONG, "The number of states in the Magnetization object and the internal value do not match. Expected %lld. Got %lld.",
                     LLD(num_states), LLD(Magnetization.NumStates()));
        return 0;
    }
    /** Every stored cell must lie inside the (sector -> sector+OpType) block it claims: by construction an entry can
        only be placed inside such a block (SectorMat::set returns PETSC_ERR_ARG_OUTOFRANGE otherwise), so this verifies
        the sector table and the cell rectangles. */
    PetscErrorCode MatCheckOperatorBlocks(const Op_t& OpType, const Mat& matin) const
    {
        PetscErrorCode ierr = CheckSectors(); CHKERRQ(ierr);
        if (!matin) SETERRQ(mpi_comm, PETSC_ERR_ARG_CORRUPT, "Matrix not yet created.");
        const dmrgx_host::SectorMat& m = matin->transpose_of ? *matin->transpose_of : *matin;
        const PetscInt shift = matin->transpose_of ? -(PetscInt)OpType : (PetscInt)OpType;
        if (m.N() != Magnetization.NumStates()) SETERRQ2(mpi_comm, 1, "Incorrect number of rows. Expected %lld. Got %lld.", LLD(Magnetization.NumStates()), LLD(m.N()));
        if (m.shift != shift) SETERRQ2(mpi_comm, PETSC_ERR_ARG_OUTOFRANGE, "Operator blocks have sector shift %d, expected %lld.", m.shift, LLD(shift));
        const PetscInt ns = Magnetization.NumSectors();
        for (const dmrgx_host::MatCell& c : m.cells) {
            const PetscInt qc = c.q + m.shift;
            if (c.q < 0 || c.q >= ns || qc < 0 || qc >= ns || c.r0 < 0 || c.c0 < 0 ||
                c.r0 + c.nr > Magnetization.Sizes(c.q) || c.c0 + c.nc > Magnetization.Sizes(qc))
                SETERRQ2(PETSC_COMM_SELF, PETSC_ERR_ARG_OUTOFRANGE, "Cell of sector %d out of bounds of block (%d -> ...)", c.q, c.q);
        }
        return 0;
    }
    PetscErrorCode MatOpCheckOperatorBlocks(const Op_t& OpType, const PetscInt& isite) const
    {
        if (isite >= num_sites) SETERRQ2(mpi_comm, PETSC_ERR_ARG_OUTOFRANGE, "Input isite (%lld) out of bounds [0,%lld).", LLD(isite), LLD(num_sites));
        const std::vector<Mat>* Op;
        switch (OpType) { case OpSm: Op = &SmData; break; case OpSz: Op = &SzData; break; case OpSp: Op = &SpData; break;
            default: SETERRQ(mpi_comm, PETSC_ERR_ARG_WRONG, "Incorrect operator type."); }
        if (!(*Op)[isite] && ops_pruned) return 0;
        return MatCheckOperatorBlocks(OpType, (*Op)[isite]);
    }
    PetscErrorCode CheckOperatorBlocks() const
    {
        if (!init) SETERRQ(mpi_comm, PETSC_ERR_ARG_CORRUPT, "Block not yet initialized.");
        PetscErrorCode ierr = CheckOperators(); CHKERRQ(ierr);
        for (PetscInt i = 0; i < num_sites; ++i) { ierr = MatOpCheckOperatorBlocks(OpSz, i); CHKERRQ(ierr); }
        for (PetscInt i = 0; i < num_sites; ++i) { ierr = MatOpCheckOperatorBlocks(OpSp, i); CHKERRQ(ierr); }
        return 0;
    }
    PetscErrorCode AssembleOperators() { return 0; }

    PetscBool HasSm() const { return init_Sm; }
    /** Sm(i) = Sp(i)^T as a view: the kernels read Sp transposed, nothing is copied. */
    PetscErrorCode CreateSm()
    {
        if (init_Sm) SETERRQ(mpi_comm, 1, "Sm was previously initialized. Call DestroySm() first.");
        PetscErrorCode ierr = CheckOperatorArray(OpSp); CHKERRQ(ierr);
        for (PetscInt i = 0; i < num_sites; ++i) {
            if (!SpData[i]) { SmData[i] = nullptr; continue; }         /* pruned site */
            auto v = std::make_shared<dmrgx_host::SectorMat>();
            v->transpose_of = SpData[i]; v->shift = OpSm; v->sizes = SpData[i]->sizes;
            SmData[i] = v;
        }
        init_Sm = PETSC_TRUE;
        return 0;
    }
    PetscErrorCode DestroySm()
    {
        if (!init_Sm && !init) return 0;
        if (!init_Sm) SETERRQ1(mpi_comm, 1, "%s was called but Sm was not yet initialized. ", __FUNCTION__);
        for (PetscInt i = 0; i < num_sites; ++i) SmData[i] = nullptr;
        init_Sm = PETSC_FALSE;
        return 0;
    }

    /** Engine extension (the reference spills whole blocks to disk instead, src/DMRGBlock.cpp:1090-1103): releases the
        Sz(i)/Sp(i) of every site with keep_sites[i] == 0.  A later access to a released operator fails loudly (null
        handle: "Term refers to an operator that does not exist", "operator not resident"), it is never silently zero. */
    PetscErrorCode PruneOperators(const std::vector<char>& keep_sites)
    {
        if (!init) return 0;
        for (PetscInt i = 0; i < num_sites; ++i) {
            if ((size_t)i < keep_sites.size() && keep_sites[(size_t)i]) continue;
            if (SzData[i] || SpData[i]) ops_pruned = PETSC_TRUE;
            SzData[i] = nullptr; SpData[i] = nullptr;         /* cells are shared_ptr views: the buffers go when the last view goes */
            if (init_Sm) SmData[i] = nullptr;
        }
        return CompactStorage();
    }
    /** The rotated operators of a block are views of ONE device arena (RotateOperators): dropping some of them frees nothing
        while the others keep the arena alive.  When less than 3/4 of the referenced storage is still in use, the surviving dense
        cells are copied into a new arena of exactly their size (one batched device copy) and the old one is released. */
    PetscErrorCode CompactStorage()
    {
        std::vector<Mat> live;
        for (PetscInt i = 0; i < num_sites; ++i) { if (SzData[i]) live.push_back(SzData[i]); if (SpData[i]) live.push_back(SpData[i]); }
        if (H) live.push_back(H);
        std::map<dmrgx_host::DevBuffer*, size_t> arena_size;
        size_t used = 0;
        for (const Mat& m : live)
            for (const dmrgx_host::MatCell& c : m->cells)
                if (c.kind == DMRGX_CELL_DENSE && c.buf) { arena_size[c.buf.get()] = c.buf->size(); used += (size_t)c.nr * (size_t)c.nc; }
        size_t held = 0;
        for (const auto& kv : arena_size) held += kv.second;
        if (held == 0 || used * 4 >= held * 3 || held * sizeof(double) < ((size_t)1 << 20)) return 0;
        std::shared_ptr<dmrgx_host::DevBuffer> arena;
        try { arena = std::make_shared<dmrgx_host::DevBuffer>(used, dmrgx_host::DevBuffer::device_only_t{}); }
        catch (const std::exception& e) { SETERRQ1(mpi_comm, PETSC_ERR_MEM, "operator storage: %s", e.what()); }
        double* base = arena->dev_uninitialised();
        if (dmrgx_memset_zero(base, used * sizeof(double), nullptr)) SETERRQ1(mpi_comm, 1, "%s", dmrgx_last_error());
        std::vector<dmrgx_axpy_task> tasks;
        std::vector<std::shared_ptr<dmrgx_host::DevBuffer>> old;         /* released when this function returns (stream-ordered pool) */
        size_t cursor = 0;
        for (const Mat& m : live)
            for (dmrgx_host::MatCell& c : m->cells) {
                if (c.kind != DMRGX_CELL_DENSE || !c.buf) continue;
                dmrgx_axpy_task t;
                t.dst = base + cursor; t.dst_base = nullptr; t.src = c.buf->dev_ro() + c.off; t.ldd = c.nc; t.lds = c.ld; t.nr = c.nr; t.nc = c.nc; t.transposed = 0; t.alpha = 1.0;
                tasks.push_back(t);
                old.push_back(c.buf);
                c.buf = arena; c.off = (int64_t)cursor; c.ld = c.nc;
                cursor += (size_t)c.nr * (size_t)c.nc;
            }
        if (!tasks.empty() && dmrgx_cells_axpy((int32_t)tasks.size(), tasks.data(), nullptr)) SETERRQ1(mpi_comm, 1, "dmrgx_cells_axpy: %s", dmrgx_last_error());
        return 0;
    }
    PetscInt NumResidentSites() const { PetscInt n = 0; for (PetscInt i = 0; i < num_sites; ++i) n += (SzData[i] && SpData[i]); return n; }
    PetscBool OpsPruned() const { return ops_pruned; }
    void SetOpsPruned() { ops_pruned = PETSC_TRUE; }
    /** A block whose sector table is known (step records) but whose operators are never computed because no later step
        of the sweep schedule reads them; any use as an input fails in CheckOperators. */
    void MarkDead() { dead = PETSC_TRUE; ops_pruned = PETSC_TRUE; }
    PetscBool Dead() const { return dead; }
    PetscInt NumRotatedOps() const { return num_rotated_ops; }

    /** this block's operators <- RotMatT . Source's operators . RotMatT^T, all of them in one device call.
        keep_sites (engine extension, default: every site): only the site operators with keep_sites[i] != 0 that Source
        holds are rotated; the others stay absent (see PruneOperators).  H is always rotated. */
    PetscErrorCode RotateOperators(SpinBase& Source, const Mat& RotMatT_in, const std::vector<char>* keep_sites = nullptr)
    {
        if (!init) SETERRQ(mpi_comm, PETSC_ERR_ARG_CORRUPT, "Block not yet initialized.");
        if (init_Sm) { PetscErrorCode ierr = DestroySm(); CHKERRQ(ierr); }
        if (!RotMatT_in || !RotMatT_in->rot) SETERRQ(mpi_comm, PETSC_ERR_ARG_WRONG, "RotMatT_in is not a rotation matrix.");
        const dmrgx_host::BasisRotation& R = *RotMatT_in->rot;
        PetscInt nrows = 0, ncols = 0;
        for (int32_t k : R.kept) nrows += k;
        for (int32_t s : R.old_sizes) ncols += s;
        if (ncols != Source.NumStates()) SETERRQ2(mpi_comm, 1, "RotMatT_in incorrect number of cols. Expected %lld. Got %lld.", LLD(Source.NumStates()), LLD(ncols));
        if (nrows != num_states) SETERRQ2(mpi_comm, 1, "RotMatT_in incorrect number of rows. Expected %lld. Got %lld.", LLD(num_states), LLD(nrows));
        if (Source.NumSites() != num_sites) SETERRQ2(mpi_comm, 1, "RotMatT_in incorrect number of sites. Expected %lld. Got %lld.", LLD(num_sites), LLD(Source.NumSites()));
        const int32_t nn = (int32_t)R.kept.size();
        std::vector<int32_t> new_of_old(R.old_sizes.size(), -1);
        for (int32_t a = 0; a < nn; ++a) new_of_old[R.old_sector[a]] = a;

        std::vector<Mat> src; std::vector<Mat> dst;
        std::vector<PetscInt> src_site;                 /* site of src[2j], src[2j+1] */
        for (PetscInt i = 0; i < num_sites; ++i) {
            const bool want = !keep_sites || ((size_t)i < keep_sites->size() && (*keep_sites)[(size_t)i]);
            if (!want || !Source.SpData[i] || !Source.SzData[i]) { ops_pruned = PETSC_TRUE; continue; }
            src.push_back(Source.SpData[i]); src.push_back(Source.SzData[i]); src_site.push_back(i);
        }
        src.push_back(Source.H);
        const size_t nops = src.size();
        std::vector<dmrgx_secop> ops(nops);
        std::vector<std::vector<dmrgx_cell>> cellstore(nops);
        std::vector<std::vector<double*>> dptr(nops, std::vector<double*>((size_t)nn, nullptr));
        std::vector<double* const*> dpp(nops);
        const std::vector<int32_t> new_sizes(R.kept.begin(), R.kept.end());
        // all rotated cells of all operators live in ONE device allocation (cells are views: buf + off), written by the
        // rotation kernels -- no per-cell hipMalloc, no host mirror
        size_t total = 0;
        for (size_t o = 0; o < nops; ++o) {
            if (!src[o]) SETERRQ1(mpi_comm, PETSC_ERR_ARG_CORRUPT, "Source operator %zu not created.", o);
            for (int32_t a = 0; a < nn; ++a) {
                const int32_t qc = R.old_sector[a] + src[o]->shift;
                if (qc < 0 || qc >= (int32_t)R.old_sizes.size() || new_of_old[qc] < 0) continue;
                total += (size_t)R.kept[a] * (size_t)R.kept[new_of_old[qc]];
            }
        }
        std::shared_ptr<dmrgx_host::DevBuffer> arena;
        try { arena = std::make_shared<dmrgx_host::DevBuffer>(total, dmrgx_host::DevBuffer::device_only_t{}); }
        catch (const std::exception& e) { SETERRQ1(mpi_comm, PETSC_ERR_MEM, "rotated operator storage: %s", e.what()); }
        double* const abase = arena->dev_uninitialised();
        size_t cursor = 0;
        for (size_t o = 0; o < nops; ++o) {
            src[o]->to_secop(ops[o], cellstore[o]);
            Mat d = std::make_shared<dmrgx_host::SectorMat>();
            d->shift = src[o]->shift; d->sizes = new_sizes;
            for (int32_t a = 0; a < nn; ++a) {
                const int32_t qc = R.old_sector[a] + d->shift;
                if (qc < 0 || qc >= (int32_t)R.old_sizes.size() || new_of_old[qc] < 0) continue;
                const int32_t ap = new_of_old[qc];
                dmrgx_host::MatCell c;
                c.q = a; c.nr = R.kept[a]; c.nc = R.kept[ap]; c.ld = c.nc;
                c.buf = arena; c.off = (int64_t)cursor;
                dptr[o][a] = abase + cursor;
                cursor += (size_t)c.nr * c.nc;
                d->cells.push_back(c);
            }
            dpp[o] = dptr[o].data();
            dst.push_back(d);
        }
        dmrgx_sectors olds{(int32_t)R.old_sizes.size(), R.old_sizes.data()};
        std::vector<const double*> rts((size_t)nn);
        for (int32_t a = 0; a < nn; ++a) rts[a] = R.rt[a]->dev_ro();
        dmrgx_rotation rot{nn, R.old_sector.data(), R.kept.data(), rts.data()};
        if (dmrgx_rotate_ops(&olds, &rot, (int32_t)nops, ops.data(), dpp.data(), nullptr)) SETERRQ1(mpi_comm, 1, "dmrgx_rotate_ops: %s", dmrgx_last_error());
        for (size_t j = 0; j < src_site.size(); ++j) { SpData[(size_t)src_site[j]] = dst[2 * j]; SzData[(size_t)src_site[j]] = dst[2 * j + 1]; }
        H = dst[nops - 1];
        num_rotated_ops = (PetscInt)nops;
        PetscErrorCode ierr = CheckOperatorBlocks(); CHKERRQ(ierr);
        return SaveAndDestroy();
    }

protected:
    /** Sz = diag(+1/2,-1/2) (spin 1/2) or diag(1,0,-1) (spin 1) */
    virtual PetscErrorCode MatSpinSzCreate(Mat& Sz)
    {
        Sz = dmrgx_host::SectorMat::Dense(OpSz, Magnetization.Sizes32());
        if (spin_type == SpinOneHalf) { Sz->set(0, 0, +0.5); Sz->set(1, 1, -0.5); }
        else { Sz->set(0, 0, +1.0); Sz->set(2, 2, -1.0); }
        return 0;
    }
    /** Sp = |0><1| (spin 1/2) or sqrt(2)(|0><1| + |1><2|) (spin 1) */
    virtual PetscErrorCode MatSpinSpCreate(Mat& Sp)
    {
        Sp = dmrgx_host::SectorMat::Dense(OpSp, Magnetization.Sizes32());
        if (spin_type == SpinOneHalf) Sp->set(0, 1, +1.0);
        else { const double s2 = 1.4142135623730951; Sp->set(0, 1, s2); Sp->set(1, 2, s2); }
        return 0;
    }

    MPI_Comm mpi_comm = PETSC_COMM_SELF;
    PetscMPIInt mpi_rank = 0, mpi_size = 1;
    PetscBool mpi_init = PETSC_FALSE, init = PETSC_FALSE, init_once = PETSC_FALSE, init_Sm = PETSC_FALSE;
    PetscBool verbose = PETSC_FALSE, saved = PETSC_FALSE, init_save = PETSC_FALSE, disk_set = PETSC_FALSE;
    PetscBool ops_pruned = PETSC_FALSE, dead = PETSC_FALSE;
    PetscInt num_rotated_ops = 0;
    PetscInt num_sites = 0, num_states = 0;
    Spin_t spin_type = SpinOneHalf;
    PetscInt _loc_dim = 2;
    std::vector<PetscScalar> _loc_qn_list = {+0.5, -0.5};
    std::vector<PetscInt> _loc_qn_size = {1, 1};
    std::vector<Mat> SzData, SpData, SmData;
    std::string save_dir, read_dir, write_dir;
};

}  // namespace Block

#endif
