// Grouped f64 GEMM on MFMA (v_mfma_f64_16x16x4_f64): task-table driven, one launch for many ragged products.
//
//   C_g[M x N] (=|+=) sum_p  A_p[M x K_p] * B_p[K_p x N]   +   sum_q alpha_q * S_q[M x N]
//
// Every operand is row-major f64 in HBM.  This is the single compute primitive behind the superblock
// MatMult (both stages), the reduced-density-matrix build and the rotation GEMMs.
#pragma once
#include "common.h"

namespace dmrgx {

enum : int32_t { GPROD_GEMM = 0, GPROD_AXPY = 1 };

struct GProd {          // one accumulation into a group's output
    const double* A;    // GEMM: M x K, row-major (k contiguous).  AXPY: unused
    const double* B;    // GEMM: K x N, row-major (n contiguous).  AXPY: source S (M x N)
    int32_t lda, ldb;
    int32_t K;
    int32_t kind;       // GPROD_GEMM | GPROD_AXPY
    double alpha;       // AXPY scale (GEMM products carry their coefficient inside A)
};

struct GGroup {         // one output matrix, tiled BM x BN
    double* C;
    int32_t ldc;
    int32_t M, N;
    int32_t prod_begin, prod_end;   // AXPY products first, then GEMM products (host sorts them)
    int32_t n_axpy;                 // number of leading AXPY products
    int32_t accumulate;             // 0: C = result, 1: C += result
};

struct GTile { int32_t group, tm, tn, pad; };

constexpr int GG_BM = 64, GG_BN = 64, GG_BK = 16, GG_THREADS = 256;

// Enqueue the tiles [0, ntiles) described by device tables.
dmrgx_status ggemm_launch(const GTile* d_tiles, const GGroup* d_groups, const GProd* d_prods, int32_t ntiles, hipStream_t st);

// Host-side helper: append the tiles of group `g` (M x N) to a tile list.
inline void ggemm_append_tiles(std::vector<GTile>& tiles, int32_t g, int32_t M, int32_t N) {
    for (int32_t tm = 0; tm < (M + GG_BM - 1) / GG_BM; ++tm)
        for (int32_t tn = 0; tn < (N + GG_BN - 1) / GG_BN; ++tn) tiles.push_back(GTile{g, tm, tn, 0});
}

}  // namespace dmrgx
