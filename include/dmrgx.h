/* dmrgx.h -- C ABI of the MI355X-native DMRG hot path (libdmrgx_hip.so).
 *
 * This is the drop-in boundary: the host sweep engine (C++, mirrors DMRGBlock / DMRGKron /
 * DMRGBlockContainer of jnvance/DMRG.x) reaches every device computation through these entry points and
 * nothing else.  Plain pointers and sizes only; no C++/torch types; every function returns a dmrgx_status
 * (0 = success), never throws, and records a message retrievable with dmrgx_last_error().
 *
 * What each entry point replaces in the reference (paths relative to the reference tree):
 *   dmrgx_kron_plan_create   <- KronBlocks_t::KronSumConstruct -> KronSumConstructShell
 *                               (src/DMRGKron.cpp:759-841, 1871-1917: term filtering is the caller's job,
 *                               operator fetch 891-989, per-row descriptors 1706-1824)
 *   dmrgx_kron_apply         <- MatMult_KronSumShell (src/DMRGKron.cpp:1827-1869), the MATOP_MULT callback
 *                               registered at src/DMRGKron.cpp:1912-1914
 *   dmrgx_kron_plan_destroy  <- MatDestroy_KronSumShell (src/DMRGKron.cpp:1919-1942)
 *   dmrgx_eigs_lowest        <- EPSSolve(EPS_HEP, EPS_SMALLEST_REAL, nev=1) as configured at
 *                               include/DMRGBlockContainer.hpp:1488-1499 [SLEPc Krylov-Schur, external]
 *   dmrgx_rdm_create         <- the device part of GetTruncation: rho_L, rho_R of every KronBlock and all their eigenpairs
 *   dmrgx_rdm_eigenvalues       (EigRDM_BlockDiag, include/DMRGBlockContainer.hpp:1715-1775, 1962-2003); the global sort and
 *   dmrgx_rdm_eigenvectors      the m-cut stay with the caller; _eigenvectors == FillRotation_BlockDiag (:2006-2057)
 *   dmrgx_rotate_ops         <- Block::SpinBase::RotateOperators (src/DMRGBlock.cpp:677-823)
 *   dmrgx_cells_axpy         <- the explicit KronSum that assembles an enlarged block's H (src/DMRGKron.cpp:612 ->
 *                               KronSumFillMatrix :1440-1446); the site operators of an enlarged block
 *                               (MatKronEyeConstruct, src/DMRGKron.cpp:52-456) are views and need no device call
 *   dmrgx_comm_*             <- the communicator of the reference's MPI path: VecScatter-to-all of x inside every MatMult
 *                               (src/DMRGKron.cpp:1833-1834) and the MPI_Allreduce behind SLEPc's VecDot / VecNorm
 *
 * Data model.  All floating point is f64 real (include/DMRGKron.hpp:395-399).  A block's basis is split in
 * Sz sectors (descending Sz); an operator with sector shift s (Op_t value: Sm=-1, Sz=0, Sp=+1,
 * include/DMRGBlock.hpp:19-27) is non-zero only in blocks (row sector q -> column sector q+s).  On the device
 * an operator is a list of *cells*: dense row-major rectangles (or scaled identities) inside those blocks.
 * The superblock vector of the target sector is the concatenation over KronBlocks k=(IL,IR) of the row-major
 * n_L(IL) x n_R(IR) matrices X_k (include/DMRGKron.hpp:160-209, 603-612).
 *
 * Threading: a plan is used from one host thread at a time; all work is enqueued on the hipStream_t passed as
 * `void* stream` (NULL = default stream).  Device pointers must come from the current HIP device.
 */
#ifndef DMRGX_H
#define DMRGX_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMRGX_ABI_VERSION 3

typedef int32_t dmrgx_status;
enum {
    DMRGX_OK = 0,
    DMRGX_ERR_ARG = 62,          /* = PETSC_ERR_ARG_WRONG: malformed descriptor                     */
    DMRGX_ERR_OUTOFRANGE = 63,   /* = PETSC_ERR_ARG_OUTOFRANGE: index/sector out of range            */
    DMRGX_ERR_MEM = 55,          /* = PETSC_ERR_MEM                                                  */
    DMRGX_ERR_DEVICE = 97,       /* HIP runtime error / no device                                    */
    DMRGX_ERR_NOTCONV = 91,      /* eigensolver did not converge within max_it                       */
    DMRGX_ERR_INTERNAL = 77
};

/* ---- library ---------------------------------------------------------------------------------------- */
int32_t      dmrgx_abi_version(void);
const char*  dmrgx_last_error(void);                 /* thread-local, valid until the next failing call  */
dmrgx_status dmrgx_device_count(int32_t* n);         /* fails loudly (DMRGX_ERR_DEVICE) without a GPU     */

/* ---- communicator: one process per GPU, collectives over RCCL / xGMI ------------------------ */
/* Replaces the reference's MPI traffic on the hot path: the VecScatter-to-all of x inside every MatMult
 * (src/DMRGKron.cpp:1833-1834), the MPI_Allreduce behind SLEPc's VecDot / VecNorm, and the rank-0 RDM solve followed by
 * a broadcast of the rotation (include/DMRGBlockContainer.hpp:1673-1677, 1812-1925).  Every collective is enqueued on the
 * caller's stream and must be called by all ranks in the same order.
 *   launch: rank 0 calls dmrgx_comm_unique_id and hands the 128 bytes to the other ranks by any means (file, socket, the
 *   launcher's store); every rank then calls dmrgx_set_device(local rank) and dmrgx_comm_init.
 * The host-staged back-end (several ranks on ONE GPU exchanging through a POSIX shared-memory segment named by rank 0)
 * exists to rehearse the N > 1 control flow on a one-GPU box; it is not a measurement path. */
typedef struct dmrgx_comm dmrgx_comm;
#define DMRGX_COMM_ID_BYTES 128
enum { DMRGX_COMM_RCCL = 1, DMRGX_COMM_HOST_STAGED = 2 };
dmrgx_status dmrgx_set_device(int32_t device);
dmrgx_status dmrgx_comm_unique_id(uint8_t* id128);
dmrgx_status dmrgx_comm_init(int32_t rank, int32_t world, const uint8_t* id128, dmrgx_comm** out);
dmrgx_status dmrgx_comm_init_host_staged(int32_t rank, int32_t world, const char* shm_name, dmrgx_comm** out);
dmrgx_status dmrgx_comm_info(const dmrgx_comm* comm, int32_t* rank, int32_t* world, int32_t* backend);
/* in place: segment r of full_vec (seg_stride doubles, this rank's = the send buffer) is replicated on every rank */
dmrgx_status dmrgx_comm_allgather(dmrgx_comm* comm, double* full_vec_dev, int64_t seg_stride, void* stream);
dmrgx_status dmrgx_comm_allreduce_sum(dmrgx_comm* comm, double* buf_dev, int64_t count, void* stream);
dmrgx_status dmrgx_comm_bcast(dmrgx_comm* comm, void* buf_dev, size_t bytes, int32_t root, void* stream);
/* host payloads (spectra, counters): recv = concatenation over ranks of bytes_per_rank bytes; synchronises the stream */
dmrgx_status dmrgx_comm_allgather_host(dmrgx_comm* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);
dmrgx_status dmrgx_comm_barrier(dmrgx_comm* comm, void* stream);      /* all ranks have finished the work queued on `stream` */
dmrgx_status dmrgx_comm_destroy(dmrgx_comm* comm);

/* ---- operator cells --------------------------------------------------------------------------------- */
enum { DMRGX_CELL_DENSE = 1, DMRGX_CELL_IDENT = 2 };

typedef struct {
    int32_t row_sector;   /* q: row sector of the (q -> q+shift) block this cell lives in               */
    int32_t r0, c0;       /* top-left corner inside that block                                          */
    int32_t nr, nc;       /* extent (IDENT: nr == nc)                                                   */
    int32_t kind;         /* DMRGX_CELL_DENSE | DMRGX_CELL_IDENT                                        */
    double  scale;        /* IDENT: the cell equals scale * I.  DENSE: ignored                          */
    const double* data;   /* DENSE: device pointer to element (r0,c0); row-major, leading dimension ld */
    int64_t ld;
} dmrgx_cell;

/* One operator of one block.  `transposed` != 0 means the operator is the transpose of the stored cells
 * (Sm(i) = Sp(i)^T, src/DMRGBlock.cpp:630-632): the cells then describe the stored operator (whose shift is
 * -shift) and are read transposed -- no Sm is ever materialised. */
typedef struct {
    int32_t shift;        /* sector shift of the operator AS USED (after the optional transpose)        */
    int32_t transposed;
    int32_t ncells;
    const dmrgx_cell* cells;   /* host array */
} dmrgx_secop;

/* Sector table of one (enlarged) block: sizes of the Sz sectors in descending-Sz order. */
typedef struct {
    int32_t nsec;
    const int32_t* size;  /* host array [nsec] */
} dmrgx_sectors;

/* ---- K1: superblock plan ---------------------------------------------------------------------------- */
/* One inter-block term a * A(left op) (x) B(right op)  (Hamiltonians::Term after the filtering/reflection of
 * src/DMRGKron.cpp:788-807).  left_op / right_op index into desc->left_ops / desc->right_ops. */
typedef struct {
    double  a;
    int32_t left_op;
    int32_t right_op;
} dmrgx_term;

typedef struct {
    dmrgx_sectors left, right;
    int32_t nblocks;               /* KronBlocks of the target sector, in the reference's order        */
    const int32_t* block_il;       /* [nblocks] left sector index  (include/DMRGKron.hpp:160-171)      */
    const int32_t* block_ir;       /* [nblocks] right sector index                                     */
    int32_t n_left_ops, n_right_ops;
    const dmrgx_secop* left_ops;   /* distinct (op,site) operators of the left block used by terms     */
    const dmrgx_secop* right_ops;
    const dmrgx_secop* h_left;     /* H_L (shift 0), term [H_L (x) 1]  (src/DMRGKron.cpp:939-944); may be NULL */
    const dmrgx_secop* h_right;    /* H_R (shift 0), term [1 (x) H_R]  (src/DMRGKron.cpp:946-951); may be NULL */
    int32_t nterms;
    const dmrgx_term* terms;
    /* striping of the right index over ranks; world_size == 1 -> plain layout */
    int32_t world_size, rank;
} dmrgx_kron_desc;

typedef struct dmrgx_kron_plan dmrgx_kron_plan;

typedef struct {
    int64_t n_states;          /* N_sb = sum_k n_L n_R                                                  */
    int64_t vec_len;           /* length of a full device vector (== n_states when world_size == 1)     */
    int64_t local_offset;      /* this rank's segment inside a full vector                              */
    int64_t local_len;         /* length of this rank's segment (incl. padding)                         */
    int64_t seg_stride;        /* stride between rank segments                                          */
    double  flops_alg;         /* algorithmic flops of ONE apply on THIS rank (SURVEY 8d F_alg)         */
    double  bytes_alg;         /* algorithmic bytes of ONE apply on THIS rank (SURVEY 8d B_alg)         */
    double  flops_exec;        /* flops the tiled kernels actually issue (padding included)             */
    double  bytes_workspace;   /* bytes of intermediates written+read per apply (not part of B_alg)     */
    int32_t n_groups;          /* merged (A,B) operator pairs                                           */
    int32_t n_tiles_stage1, n_tiles_stage2;
    int32_t n_tiles_big;       /* of those, 128x128 macro tiles (the rest are 64x64)                      */
    double  flops_alg_big;     /* part of flops_alg executed by the 128x128 kernel                        */
} dmrgx_kron_info;

dmrgx_status dmrgx_kron_plan_create(const dmrgx_kron_desc* desc, void* stream, dmrgx_kron_plan** out);
dmrgx_status dmrgx_kron_plan_info(const dmrgx_kron_plan* plan, dmrgx_kron_info* info);
/* y_local <- (H x)[this rank's segment].  x_full: full vector (vec_len), y_local: points at the start of this
 * rank's segment of a full vector or at a separate buffer of local_len doubles.  world_size==1: y = H x. */
dmrgx_status dmrgx_kron_apply(dmrgx_kron_plan* plan, const double* x_full, double* y_local, void* stream);
dmrgx_status dmrgx_kron_plan_destroy(dmrgx_kron_plan* plan);
/* d_local[e] <- <e| H |e> for the basis states of this rank's segment (local_len doubles, padding 0): the diagonal of the
 * superblock Hamiltonian, = sum over H_L (x) 1, 1 (x) H_R and the sector-diagonal terms (Sz Sz) of diag(A)[l] diag(B)[r]
 * -- what a diagonally preconditioned eigensolver (SLEPc: -H_eps_type gd, -H_st_pc_type jacobi) asks MatGetDiagonal for; the
 * reference's shell matrix does not implement it (src/DMRGKron.cpp:1912-1914 registers MATOP_MULT only). */
dmrgx_status dmrgx_kron_diag(dmrgx_kron_plan* plan, double* d_local_dev, void* stream);
/* Optional per-launch timing of the two GEMM stages with HIP events recorded on the apply's own stream
 * (the reference's analogue is the -DDMRG_KRON_TIMINGS accumulators, include/MiscTools.hpp:17-59).
 * enable != 0 resets and starts recording (up to 4096 applies), enable == 0 stops. */
dmrgx_status dmrgx_kron_plan_timing(dmrgx_kron_plan* plan, int32_t enable);
/* Synchronises the recorded events and returns the summed kernel time in milliseconds of the four GEMM launches
 * of an apply: ms[0] stage-1 128x128 tiles, ms[1] stage-1 64x64 tiles, ms[2] stage-2 128x128, ms[3] stage-2 64x64. */
dmrgx_status dmrgx_kron_plan_timing_read(dmrgx_kron_plan* plan, double* ms4, int64_t* n_applies);
/* Convert between the reference's vector layout (KronBlocks order, n_states doubles, host or device) and the
 * striped full-vector layout (identity copy when world_size == 1). */
dmrgx_status dmrgx_kron_vec_to_striped(const dmrgx_kron_plan* plan, const double* v_ref_dev, double* v_full_dev, void* stream);
dmrgx_status dmrgx_kron_vec_from_striped(const dmrgx_kron_plan* plan, const double* v_full_dev, double* v_ref_dev, void* stream);

/* Host-only helper (no device needed): columns [*c0, *c1) are stripe number `rank` of a KronBlock whose right sector has n_right
 * states cut for `world_size` ranks -- the cut rule used by every plan (cuts on whole GEMM tiles where possible). */
dmrgx_status dmrgx_stripe_bounds(int32_t n_right, int32_t world_size, int32_t rank, int32_t* c0, int32_t* c1);
/* The columns rank `rank` owns in the KronBlock number `block` (position in desc->block_il/ir): stripe (rank + block) mod world_size
 * of dmrgx_stripe_bounds -- the stripes are dealt round the ranks block by block so that the ragged last stripe moves around. */
dmrgx_status dmrgx_stripe_bounds_of_block(int32_t n_right, int32_t world_size, int32_t rank, int32_t block, int32_t* c0, int32_t* c1);

/* ---- generic grouped f64 GEMM (used by K1/K3/K6; exposed for tests) ---------------------------------- */
/* C[M x N] (row-major, ldc) = A[M x K] (row-major, lda) * B[K x N] (row-major, ldb), device pointers. */
dmrgx_status dmrgx_dgemm_nn(int32_t M, int32_t N, int32_t K, const double* A, int64_t lda,
                            const double* B, int64_t ldb, double* C, int64_t ldc, void* stream);
/* The same for `count` independent products in one grouped launch (operator products per sector block, correlators:
 * the MatMatMult calls of include/DMRGBlockContainer.hpp:2378-2395).  accumulate != 0: C += A*B.  Outputs must not
 * overlap.  Both GEMM entry points are asynchronous on `stream` (the task array itself is consumed before returning). */
typedef struct {
    int32_t M, N, K, accumulate;
    const double* A; int64_t lda;
    const double* B; int64_t ldb;
    double* C; int64_t ldc;
} dmrgx_gemm_task;
dmrgx_status dmrgx_dgemm_batch(int32_t count, const dmrgx_gemm_task* tasks, void* stream);

/* ---- K2: lowest eigenpair of the planned superblock Hamiltonian -------------------------------------- */
typedef struct {
    int32_t ncv;        /* Krylov subspace size (SLEPc default for nev=1: 16)                            */
    int32_t max_it;     /* maximum number of restarts                                                    */
    double  tol;        /* converged when ||r|| <= tol * |theta|  (SLEPc default criterion, default 1e-8) */
    uint64_t seed;      /* start vector: counter-based uniform(-1,1) stream of this seed, unless ...      */
    int32_t use_initial; /* ... use_initial != 0: psi_full holds the start vector                         */
    int32_t max_matvec;  /* > 0: stop after exactly this many MatMults (status DMRGX_ERR_NOTCONV unless converged
                            earlier); used by benchmarks to time a fixed number of Lanczos steps               */
    /* collective hooks for world_size > 1 (NULL when world_size == 1).  They must be stream-ordered on
     * `stream`: allgather(sendbuf=this rank's segment, full vector) and allreduce_sum(buf, count). */
    dmrgx_status (*allgather)(void* user, double* full_vec, int64_t seg_stride, void* stream);
    dmrgx_status (*allreduce_sum)(void* user, double* buf, int64_t count, void* stream);
    void* user;
    /* native collectives: when `comm` is set (and the hooks are NULL) the solver calls dmrgx_comm_allgather /
     * dmrgx_comm_allreduce_sum itself -- the product path; the hooks remain for harnesses that own their communicator */
    dmrgx_comm* comm;
    /* 0: thick-restart Lanczos (SLEPc's default for this solve, -H_eps_type krylovschur).  1: generalized Davidson with the
     * diagonal of H_sb as preconditioner (-H_eps_type gd with a Jacobi preconditioner): same convergence criterion (true
     * residual), about a quarter fewer MatMults from the engine's transformed start vectors; the projected problem is
     * solved on the device, the host only looks at (|r|, theta) one iteration behind the queue.  Applies when use_initial is set; from a random start the Lanczos
     * path is used (the preconditioned iteration is twice as slow there). */
    int32_t method;
    /* > 0 (with use_initial): the start vector is only trusted if its squared norm is at least this -- the engine's start vectors are
     * projections of a normalised state, so their norm says how much of it survived.  A lighter vector is dropped and the solve runs
     * from the random start vector (stats->start_rejected = 1).  The norm is read with the first coefficients that come back to the
     * host anyway: no extra synchronisation.  0: no check beyond "not zero / NaN". */
    double min_initial_norm2;
    /* method 1: number of lowest Ritz vectors the search space is restarted to, beside the previous iteration's Ritz vector (SLEPc:
     * -eps_gd_minv with -eps_gd_plusk 1).  0: the default, 1.  The search space itself holds `ncv` vectors (0: the default of the
     * method -- 16 for method 0, 8 for method 1). */
    int32_t gd_minv;
    int32_t reserved_;
} dmrgx_eigs_opts;

typedef struct {
    int32_t n_matvec;       /* number of dmrgx_kron_apply calls ("superblock MatMults")                  */
    int32_t n_restart;
    int32_t converged;
    int32_t start_rejected; /* 1: the supplied start vector was dropped (zero / NaN norm, or below min_initial_norm2)  */
    double  residual;       /* final ||H psi - e0 psi||                                                  */
    double  seconds;        /* wall time of the solve (host clock around a stream sync)                  */
} dmrgx_eigs_stats;

dmrgx_status dmrgx_eigs_lowest(dmrgx_kron_plan* plan, const dmrgx_eigs_opts* opts, double* e0,
                               double* psi_full, dmrgx_eigs_stats* stats, void* stream);

/* Distributed solves only: where the MatMults of the solves since the last reset spent their time on this rank -- the all-gather that
 * rebuilds x (the reference's VecScatter-to-all, src/DMRGKron.cpp:1833-1834) and the apply behind it -- from HIP events on the
 * solver's stream (milliseconds, summed over n_matvec MatMults).  reset != 0 clears the totals.  Any pointer may be null. */
dmrgx_status dmrgx_eigs_comm_timing(double* allgather_ms, double* apply_ms, int64_t* n_matvec, int32_t reset);

/* ---- K3/K4: reduced density matrices + full spectra ------------------------------------------------------ */
/* For every KronBlock k of the layout: rho_L = Psi Psi^T, rho_R = Psi^T Psi (Psi = n_L x n_R row-major slice of
 * psi_dev in the reference's vector layout) and ALL eigenpairs of each, largest first -- the device part of
 * GetTruncation / EigRDM_BlockDiag (include/DMRGBlockContainer.hpp:1715-1775, 1962-2003).  The global sort, the
 * m-cut and the sector bookkeeping (:1795, 1850-1892) stay with the caller. side: 0 = left (rho_L), 1 = right. */
typedef struct dmrgx_rdm dmrgx_rdm;
dmrgx_status dmrgx_rdm_create(const dmrgx_sectors* left, const dmrgx_sectors* right, int32_t nblocks,
                              const int32_t* block_il, const int32_t* block_ir, const double* psi_dev,
                              void* stream, dmrgx_rdm** out);
/* The same with a starting basis: v0_rows[2*k + side] is NULL or a device pointer to an n x n row-major ORTHOGONAL matrix whose
 * rows approximately diagonalise that block's density matrix (the rows written by dmrgx_rdm_eigenvectors(count = n) at the
 * previous visit of the same block).  Results do not depend on v0_rows, only the number of Jacobi sweeps can.  It is a
 * hint: the default solver preconditions every matrix with one Householder-QR step on the sorted matrix (csrc/hqr.hip),
 * which supersedes it; with DMRGX_RDM_QR=0 the matrix is transformed into the supplied basis before the iteration. */
/* Multi-GPU form: only the density matrices selected by side_mask[k] (bit 0: rho_L, bit 1: rho_R of KronBlock k) are built and
 * diagonalised by this rank -- the matrices of a step are dealt over the ranks by their n^3 cost; psi_dev is the
 * whole vector on every rank.  Queries for a matrix that was not selected fail with DMRGX_ERR_ARG. */
dmrgx_status dmrgx_rdm_create_subset(const dmrgx_sectors* left, const dmrgx_sectors* right, int32_t nblocks,
                                     const int32_t* block_il, const int32_t* block_ir, const double* psi_dev,
                                     const uint8_t* side_mask, void* stream, dmrgx_rdm** out);
dmrgx_status dmrgx_rdm_create_warm(const dmrgx_sectors* left, const dmrgx_sectors* right, int32_t nblocks,
                                   const int32_t* block_il, const int32_t* block_ir, const double* psi_dev,
                                   const double* const* v0_rows, void* stream, dmrgx_rdm** out);
/* host_out[0..n) = eigenvalues of block k's matrix, descending (n = sector size on that side). */
dmrgx_status dmrgx_rdm_eigenvalues(const dmrgx_rdm* rdm, int32_t side, int32_t k, double* host_out);
/* dst_dev[r*ld + i], r < count: the eigenvector of the r-th largest eigenvalue as a ROW (a row of RotMatT,
 * == FillRotation_BlockDiag, include/DMRGBlockContainer.hpp:2032-2054). */
dmrgx_status dmrgx_rdm_eigenvectors(const dmrgx_rdm* rdm, int32_t side, int32_t k, int32_t count, double* dst_dev, int64_t ld, void* stream);
/* The same for several (density matrix, destination) pairs in one launch (the rotation of a truncation step asks for one block of rows
 * per kept sector: a dozen launches of a few microseconds each at m = 512). */
typedef struct { int32_t side, k, count, pad; double* dst_dev; int64_t ld; } dmrgx_rdm_vec_task;
dmrgx_status dmrgx_rdm_eigenvectors_batch(const dmrgx_rdm* rdm, int32_t ntasks, const dmrgx_rdm_vec_task* tasks, void* stream);
/* Optional second phase between the spectra and the eigenvectors.  dmrgx_rdm_create* returns with every spectrum final but -- with the direct
 * solver -- no eigenvector formed: the m-cut of GetTruncation (include/DMRGBlockContainer.hpp:1795-1875) is taken on the spectra, and only
 * then are the eigenvectors of the KEPT states computed: counts[2*k + side] = number of (largest) eigenvalues of that density matrix whose
 * eigenvectors are wanted (entries of matrices this rank did not build are ignored).  The last merge of the divide and conquer, the
 * back-transformation and the verification then run on half-width matrices when half of the states are kept.  The spectra do not change
 * (dmrgx_rdm_eigenvalues: the solver's own eigenvalues, descending); dmrgx_rdm_eigenvectors serves count <= counts[..].  A caller that never
 * selects gets every eigenvector at its first dmrgx_rdm_eigenvectors call.  `psi_dev` of the create call must stay unchanged until then.
 * Verification: every selected eigenvalue is computed a second time as the Rayleigh quotient |Psi^T u|^2 of its finished eigenvector; the two
 * are compared when the object is destroyed -- dmrgx_rdm_destroy returns DMRGX_ERR_NOTCONV if they disagree (the counterpart of the
 * reference's "all eigenpairs converged" check, include/DMRGBlockContainer.hpp:1987) -- so the comparison costs no synchronisation. */
dmrgx_status dmrgx_rdm_select(dmrgx_rdm* rdm, const int32_t* counts, void* stream);
/* What the solver of this set of density matrices did.  The reference checks "all eigenpairs converged" after every LAPACK call
 * (include/DMRGBlockContainer.hpp:1987); here a failure would not be a wrong result but a silently slower path, so the path is reported:
 * DMRGRun.json counts TridFallbacks / TridLaunchPathCalls from it and bench.py refuses a leg that took an unexpected path. */
typedef struct {
    int32_t n_sweeps;                   /* block-Jacobi solver: outer sweeps (0 for the direct solver)                                   */
    int32_t solver;                     /* 0: Householder tridiagonalisation + divide and conquer (csrc/symeig.hip), 1: block Jacobi    */
    int32_t trid_persistent_matrices;   /* matrices tridiagonalised by the persistent LDS-resident kernel                                */
    int32_t trid_launch_matrices;       /* ... by one launch per column: by design (order > ~1700) or after a time-out                   */
    int32_t max_workgroups_per_matrix;  /* persistent kernel: workgroups that shared the rows of one matrix                              */
    int32_t merge_levels;               /* depth of the divide-and-conquer tree of the largest matrix                                    */
    int32_t wy_blocks_max;              /* compact-WY blocks (64 reflectors) in the back-transformation of the largest matrix            */
    int32_t timed_out;                  /* this call's persistent round timed out (1) / was lapped (2) and was repeated by launches      */
    int32_t process_timeouts;           /* persistent rounds that timed out in this process so far                                       */
    int32_t persistent_off;             /* the persistent kernel is switched off for the rest of the process (time-out, shared GPU)      */
} dmrgx_rdm_report;
dmrgx_status dmrgx_rdm_info(const dmrgx_rdm* rdm, dmrgx_rdm_report* out);
dmrgx_status dmrgx_rdm_destroy(dmrgx_rdm* rdm);

/* ---- K6: operator rotation + dense-cell accumulate ------------------------------------------------------------ */
/* dst[i*ldd + j] += alpha * src(i,j), src(i,j) = src[i*lds + j] or (transposed) src[j*lds + i]; nr x nc is the shape of
 * the DESTINATION rectangle.  Tasks whose destinations may overlap must carry the same dst_base (e.g. the owning
 * sector block): they are then applied in submission order; dst_base == NULL means "no overlap with other tasks".
 * src == NULL adds alpha to the diagonal of a square destination (scaled-identity source cell). */
typedef struct {
    double* dst; const double* dst_base; const double* src;
    int64_t ldd, lds; int32_t nr, nc; int32_t transposed; double alpha;
} dmrgx_axpy_task;
dmrgx_status dmrgx_cells_axpy(int32_t ntasks, const dmrgx_axpy_task* tasks, void* stream);

/* Truncation as a block rotation: new sector a keeps kept[a] states of old sector old_sector[a] (ascending), rows of
 * RotMatT for it are rot_t[a] (kept[a] x n_old, row-major, ld = n_old; device) -- include/DMRGBlockContainer.hpp:2032-2054. */
typedef struct {
    int32_t n_new;
    const int32_t* old_sector;
    const int32_t* kept;
    const double* const* rot_t;
} dmrgx_rotation;
/* For every source operator o (cells in the OLD sector basis): dst_blocks[o][a] <- RT_a . O_{q -> q+shift} . RT_a'^T with
 * q = old_sector[a] and a' the new sector cut from old sector q+shift; dst_blocks[o][a] is a caller-allocated dense
 * kept[a] x kept[a'] row-major block (ignored / may be NULL when a' does not exist).  == RotateOperators
 * (src/DMRGBlock.cpp:763-772) for Sz(i), Sp(i) and H at once. */
dmrgx_status dmrgx_rotate_ops(const dmrgx_sectors* old_sectors, const dmrgx_rotation* rot, int32_t nops,
                              const dmrgx_secop* src_ops, double* const* const* dst_blocks, void* stream);

/* ---- device memory (so that the host engine needs no HIP headers) ------------------------------------------ */
/* Blocks come from a size-class pool inside the library: dmrgx_free never synchronises the device, and a freed block is
 * recycled in STREAM ORDER -- work already queued on it may still be running, the next owner's operations are queued
 * behind it on the same stream.  A caller that frees on one stream and reuses on another calls dmrgx_stream_sync in
 * between (the engine and the Python wrappers use a single stream).  DMRGX_POOL=0 makes every free a plain hipFree. */
dmrgx_status dmrgx_malloc(void** dev_ptr, size_t bytes);
dmrgx_status dmrgx_free(void* dev_ptr);
/* device bytes handed out by the pool now, cached for reuse, and the high-water mark of the first (what the reference's
 * disk spill bounds: src/DMRGBlock.cpp:1090-1103); any pointer may be NULL */
dmrgx_status dmrgx_mem_stats(size_t* in_use, size_t* cached, size_t* peak_in_use);
dmrgx_status dmrgx_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
dmrgx_status dmrgx_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);   /* synchronises the stream */
dmrgx_status dmrgx_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes, void* stream);
dmrgx_status dmrgx_memset_zero(void* dst_dev, size_t bytes, void* stream);
dmrgx_status dmrgx_stream_sync(void* stream);
/* *host_out = <x, y> (device vectors, fixed summation order); replaces VecDot in the correlator path
 * (include/DMRGBlockContainer.hpp:2287-2293).  Synchronises the stream. */
dmrgx_status dmrgx_dot(int64_t n, const double* x_dev, const double* y_dev, double* host_out, void* stream);
/* The same without the synchronisation: the sum (same summation order) is written to dev_out[0] in stream order -- lets a
 * caller queue many expectation values and fetch them with one dmrgx_memcpy_d2h. */
dmrgx_status dmrgx_dot_async(int64_t n, const double* x_dev, const double* y_dev, double* dev_out, void* stream);
/* Many 2-D (Frobenius) inner products in one launch: dev_out[t.out] = sum over the tasks with that `out` of
 * sum_ij a[i*lda + j] * b[i*ldb + j]; outputs that no task names are left untouched.  Fixed summation order.  With the
 * Gram blocks G_k = X_k X_k^T of the state this evaluates <psi|P (x) 1|psi> = sum_k <P[IL(k)], G_k> for a whole table of
 * system-block correlators at once (the MatMult + VecDot pairs of include/DMRGBlockContainer.hpp:2287-2293). */
typedef struct { const double* a; int64_t lda; const double* b; int64_t ldb; int32_t nr, nc, out, pad; } dmrgx_dot2d_task;
dmrgx_status dmrgx_dot2d_batch(int32_t count, const dmrgx_dot2d_task* tasks, double* dev_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DMRGX_H */
