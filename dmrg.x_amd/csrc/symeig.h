// Batched symmetric eigensolver (f64) for the reduced density matrices: Householder tridiagonalisation, tridiagonal divide and
// conquer, blocked back-transformation (symeig.hip).  Replaces the QR-preconditioned block-Jacobi iteration of rounds 1-2
// (rdm.hip keeps it behind DMRGX_RDM_SOLVER=jacobi and for matrices above SYMEIG_MAX_N).
//
// Takes the place of the reference's EPSLAPACK call in EigRDM_BlockDiag (include/DMRGBlockContainer.hpp:1976-1982: all
// eigenpairs of one dense symmetric block): same mathematical result (eigenvalues to c n eps ||A||, eigenvectors orthogonal to
// round-off), computed for every block of a truncation step at once.
#pragma once
#include "common.h"

namespace dmrgx {

struct SymEigMat {
    int32_t n;       // order (0 allowed: nothing is done)
    int32_t lda;     // leading dimension of A (row-major)
    int32_t ldx;     // leading dimension of X
    int32_t pad;
    double* A;       // in: symmetric n x n, BOTH triangles stored; overwritten
    double* X;       // out: eigenvectors as COLUMNS of the row-major n x n array (column c belongs to w[c])
    double* w;       // out (device): eigenvalues, ascending
};

constexpr int SYMEIG_MAX_N = 3072;     // the merge kernels keep O(n) vectors of one sub-problem in LDS

// Enqueue the eigendecomposition of every matrix on `st`; workspace comes from the stream-ordered pool.  The results are only valid
// after `st` has been synchronised by the caller -- but the call is NOT free of host synchronisation: when the persistent
// tridiagonalisation ran, the call waits for it once (it reads the kernel's status word to decide between carrying on and repeating
// the step by launches) before it enqueues the rest, and the WY factors run on an internal second stream that joins `st` again
// before the call returns.
// What a call did (optional): which tridiagonalisation path the matrices took, how the persistent kernel was laid out, how deep the
// divide-and-conquer tree went -- the reference checks "all eigenpairs converged" after its LAPACK call
// (include/DMRGBlockContainer.hpp:1987); here the equivalent of a silent failure would be a silently slower path.
struct SymEigReport {
    int32_t persistent_matrices = 0;     // tridiagonalised by the persistent LDS-resident kernel
    int32_t launch_matrices = 0;         // tridiagonalised by one launch per column: by design (rows do not fit) or after a time-out
    int32_t max_workgroups_per_matrix = 0;
    int32_t merge_levels = 0;            // depth of the divide-and-conquer tree of the largest matrix
    int32_t wy_blocks_max = 0;           // compact-WY blocks of the back-transformation of the largest matrix
    int32_t timed_out = 0;               // this call's persistent round ran into its bounded spin (1) or was lapped (2) and was repeated by launches
};
// Two-phase use (round 5): a truncation keeps about half of the eigenvectors, and which ones is only known once every spectrum has been seen
// on the host.  With `deferred` the call stops behind the last level's secular equations -- the eigenvalues w are final, the eigenvectors are
// not formed: the last merge's GEMM and the whole back-transformation are left to symeig_finish, which runs them for the `keep[i]` LARGEST
// eigenvalues of matrix i only (the last keep[i] columns of X; the other columns of X are then undefined).  The object owns the workspace
// in between and must be finished (or dropped) on the same stream.
struct SymEigDeferred {
    struct Mat { SymEigMat m; int64_t VT, Vc, TV, Q1, U, W; int32_t nblk, depth; };
    std::vector<Mat> mats;               // the matrices of order > 0, in the order of the call
    DevBuf dbuf, ibuf;                   // workspace of the call
    bool pending = false;
    const int32_t* status_host = nullptr;   // (pinned) status word of the persistent rounds, valid behind the caller's next synchronisation of the stream
};
// After the caller has synchronised the stream: 0, or the status (1: time-out, 2: lapped) of a deferred call's persistent rounds -- its results are
// then garbage: note the event (symeig_note_timeout: message, the persistent kernel is switched off for the process) and repeat the call.
int32_t symeig_deferred_timed_out(const SymEigDeferred& d);
void symeig_note_timeout(int32_t status);
dmrgx_status symeig_batched(const std::vector<SymEigMat>& mats, hipStream_t st, SymEigReport* report = nullptr, SymEigDeferred* deferred = nullptr);
dmrgx_status symeig_finish(SymEigDeferred& d, const std::vector<int32_t>& keep, hipStream_t st);

// The tridiagonalisation normally runs as ONE persistent launch with the matrices resident in the LDS of most CUs of the
// chip; processes that share a GPU with other ranks (the host-staged rehearsal communicator) switch it off and use one launch per
// column (also selected by DMRGX_TRID=launch, and automatically after a bounded-spin timeout).
void symeig_set_persistent(bool on);
// process-wide: persistent rounds that timed out so far, and whether the persistent kernel is switched off (by a time-out or by the call above)
void symeig_process_state(int32_t* timeouts, int32_t* persistent_off);

}  // namespace dmrgx
