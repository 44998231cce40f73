// Batched Householder QR with explicit Q^T (f64), the preconditioner of the density-matrix eigensolver (rdm.hip).
//
// For each matrix the caller lays out the n x 2n row-major work array  B = [ A | I ]  in the arena; on return the right
// half holds Q^T (A = Q R; the left half is scratch).  Orthogonality of Q is at round-off whatever the conditioning
// of A (Householder reflectors, compact-WY panels of width 32), which is what makes it usable on density matrices whose
// spectrum spans 30 decades.
#pragma once
#include "common.h"

namespace dmrgx {

struct HqrMat {
    int64_t b_off;      // B: n x (2n) row-major, arena offset in doubles
    int64_t v_off;      // V scratch: n x 32
    int64_t t_off;      // T scratch: 32 x 32
    int32_t n, pad;
};

constexpr int HQR_MAX_N = 15360;       // panel kernel keeps one reflector (n doubles) in LDS

// Enqueue the factorisation of all matrices (d_mats: the same table on the device).  Asynchronous on `st`.
dmrgx_status hqr_batched(const std::vector<HqrMat>& mats, const HqrMat* d_mats, double* buf, hipStream_t st);

}  // namespace dmrgx
