// Grouped f64 GEMM on MFMA (v_mfma_f64_16x16x4_f64): task-table driven, one launch for many ragged products.
//
//   C_g[M x N] (=|+=) sum_p  A_p[M x K_p] * B_p[K_p x N]   +   sum_q alpha_q * S_q[M x N]
//
// Every operand is row-major f64 in HBM.  This is the single compute primitive behind the superblock
// MatMult (both stages), the reduced-density-matrix build and the rotation GEMMs.
#pragma once
#include "common.h"
#include <algorithm>

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

struct GTile { int32_t group, tm, tn, pad; };   // pad: host, before ggemm_schedule: cost in k-steps; device: first GEMM product of the group

constexpr int GG_BM = 64, GG_BN = 64, GG_BK = 16, GG_THREADS = 256;

// Enqueue the scheduled tile list [0, ntiles) described by device tables.  big != 0: every tile is a 128 x 128 macro tile
// (coordinates still in 64-units), else 64 x 64.
dmrgx_status ggemm_launch(const GTile* d_tiles, const GGroup* d_groups, const GProd* d_prods, int32_t ntiles, hipStream_t st, int big = 0);

// Host-side helper: append the tiles of group `g` (M x N) to a tile list in 8 x 8 clusters (tiles of a cluster share
// A row-panels and B column-panels; the scheduler keeps a cluster on one XCD so they meet in its L2); `cost` (k-steps of
// the group's product list) is stored in GTile::pad for the scheduler.
constexpr int GG_CLUSTER = 8;   // cluster edge in tiles (4 and 16 measured within 1-2 %, rounds 2 and 4)
inline void ggemm_append_tiles(std::vector<GTile>& tiles, int32_t g, int32_t M, int32_t N, int32_t cost = 1) {
    const int32_t TM = (M + GG_BM - 1) / GG_BM, TN = (N + GG_BN - 1) / GG_BN;
    for (int32_t bm = 0; bm < TM; bm += GG_CLUSTER)
        for (int32_t bn = 0; bn < TN; bn += GG_CLUSTER)
            for (int32_t tm = bm; tm < std::min(TM, bm + GG_CLUSTER); ++tm)
                for (int32_t tn = bn; tn < std::min(TN, bn + GG_CLUSTER); ++tn) tiles.push_back(GTile{g, tm, tn, cost});
}

inline void ggemm_append_tiles_mixed(std::vector<GTile>& big, std::vector<GTile>& small, int32_t g, int32_t M, int32_t N, int32_t cost = 1, bool allow_big = true) {
    const int32_t mb = allow_big ? (M / 128) * 2 : 0, nb = allow_big ? (N / 128) * 2 : 0;   // core extent in 64-units
    for (int32_t bm = 0; bm < mb; bm += 2 * GG_CLUSTER)
        for (int32_t bn = 0; bn < nb; bn += 2 * GG_CLUSTER)
            for (int32_t tm = bm; tm < std::min(mb, bm + 2 * GG_CLUSTER); tm += 2)
                for (int32_t tn = bn; tn < std::min(nb, bn + 2 * GG_CLUSTER); tn += 2) big.push_back(GTile{g, tm, tn, 4 * cost});
    const int32_t TM = (M + GG_BM - 1) / GG_BM, TN = (N + GG_BN - 1) / GG_BN;
    for (int32_t bm = 0; bm < TM; bm += GG_CLUSTER)
        for (int32_t bn = 0; bn < TN; bn += GG_CLUSTER)
            for (int32_t tm = bm; tm < std::min(TM, bm + GG_CLUSTER); ++tm)
                for (int32_t tn = bn; tn < std::min(TN, bn + GG_CLUSTER); ++tn)
                    if (tm >= mb || tn >= nb) small.push_back(GTile{g, tm, tn, cost});
}

// XCD-aware, cost-balanced launch order.  Blocks b, b+8, b+16.. run on one XCD (each XCD has its own L2), so the
// list is rebuilt as 8 interleaved per-XCD lists: clusters of tiles that share an A row-panel (same group and
// tile row) stay on one XCD for L2 reuse, clusters are dealt longest-first to the least-loaded XCD (LPT), and each
// XCD runs its longest clusters first so the tail is made of short tiles.  Lists are padded with group = -1.
//
// Since round 4 the launch keeps `ggemm_slots` RESIDENT workgroups (every workgroup slot of the chip): workgroup b starts on entry b of
// the scheduled list and then claims the next unclaimed entry of its XCD's queue (entries x, x + 8, x + 16, ... belong to XCD x) with
// one atomic per tile, so a queue is worked off in order by whichever workgroup of that XCD is free.  On the device GTile::pad is the
// index of the group's first GEMM product -- -1 if it has none -- (on the host, before scheduling, the tile's cost in k-steps), which is
// why the scheduler needs the groups.
// EVERY list handed to ggemm_launch must have gone through ggemm_schedule.
int ggemm_slots(int unit = 1);
void ggemm_schedule_core(std::vector<GTile>& tiles, const std::vector<int32_t>& first_gemm_product_of_group, int unit);
template <class GroupVec>
inline void ggemm_schedule(std::vector<GTile>& tiles, const GroupVec& groups, int unit = 1) {   // unit = 2 for lists of 128 x 128 tiles
    std::vector<int32_t> p0(groups.size());
    for (size_t i = 0; i < groups.size(); ++i) p0[i] = groups[i].prod_begin + groups[i].n_axpy < groups[i].prod_end ? groups[i].prod_begin + groups[i].n_axpy : -1;
    ggemm_schedule_core(tiles, p0, unit);
}

}  // namespace dmrgx
