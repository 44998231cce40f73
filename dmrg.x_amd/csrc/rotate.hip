// K6: basis rotation of block operators, and the dense-cell accumulate primitive used to assemble an enlarged
// block's Hamiltonian.
//
// dmrgx_rotate_ops replaces Block::SpinBase::RotateOperators (reference src/DMRGBlock.cpp:677-823): for every
// operator O in {Sz(i), Sp(i), H} it forms O' = RotMatT . O . RotMat (MatMatMatMult, :766-771).  RotMatT is
// sector-block sparse (rows of new sector a only touch columns of the old sector it was cut from,
// include/DMRGBlockContainer.hpp:2042-2050), so per sector pair O'_{a,a'} = RT_a O_{q,q'} RT_{a'}^T.  All operators
// and sector pairs of a block go through TWO launches of the grouped MFMA-f64 GEMM:
//     stage A:  W_c   = RT_a[:, rows of cell c] . cell_c                (identity cells: scaled copy)
//     stage B:  O'_aa' = sum_c W_c . UT_a'[cols of cell c, :]           (UT = RT^T, transposed once per call)
// so the structural zeros of an enlarged block's O (x) 1_2 operators are never multiplied.
//
// dmrgx_cells_axpy: dst_cell (+)= alpha * src_cell (optionally transposed) for a batch of dense rectangles -- the
// device form of the explicit KronSum that builds an enlarged block's H (reference src/DMRGKron.cpp:612 ->
// KronSumFillMatrix :1440-1446 restricted to  H_old (x) 1 + sum_t a_t O_i (x) s_site).
#include "ggemm.h"
#include <chrono>
#include <algorithm>
#include <map>

namespace dmrgx {
namespace {

struct AxTask { double* dst; const double* src; int64_t ldd, lds; int32_t nr, nc, tr, pad; double alpha; };
struct AxTile { int32_t task, ti, tj, pad; };

__global__ void __launch_bounds__(256) cells_axpy_kernel(const AxTile* __restrict__ tiles, const AxTask* __restrict__ tasks)
{
    __shared__ double buf[32][33];
    const AxTile t = tiles[blockIdx.x];
    const AxTask k = tasks[t.task];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i0 = t.ti * 32, j0 = t.tj * 32;
    if (!k.src) {                       // identity source: dst += alpha * I
        if (t.ti == t.tj)
            for (int r = ty; r < 32; r += 8) {
                const int i = i0 + r, j = j0 + tx;
                if (i == j && i < k.nr && j < k.nc) k.dst[(size_t)i * k.ldd + j] += k.alpha;
            }
    } else if (!k.tr) {
        for (int r = ty; r < 32; r += 8) {
            const int i = i0 + r, j = j0 + tx;
            if (i < k.nr && j < k.nc) k.dst[(size_t)i * k.ldd + j] += k.alpha * k.src[(size_t)i * k.lds + j];
        }
    } else {
        for (int r = ty; r < 32; r += 8) {
            const int j = j0 + r, i = i0 + tx;
            buf[r][tx] = (i < k.nr && j < k.nc) ? k.src[(size_t)j * k.lds + i] : 0.0;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int i = i0 + r, j = j0 + tx;
            if (i < k.nr && j < k.nc) k.dst[(size_t)i * k.ldd + j] += k.alpha * buf[tx][r];
        }
    }
}

}  // namespace
}  // namespace dmrgx

using namespace dmrgx;

extern "C" dmrgx_status dmrgx_cells_axpy(int32_t n, const dmrgx_axpy_task* tasks, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || (n > 0 && !tasks)) DMRGX_FAIL(DMRGX_ERR_ARG, "cells_axpy: bad argument");
    if (n == 0) return DMRGX_OK;
    // tasks may target the same destination: run them in submission order, one launch per "round" of writers
    std::map<const double*, int32_t> rounds;
    std::vector<AxTask> ht(n);
    std::vector<int32_t> rnd(n);
    int32_t max_round = 0;
    for (int32_t i = 0; i < n; ++i) {
        const dmrgx_axpy_task& t = tasks[i];
        if (!t.dst || t.nr <= 0 || t.nc <= 0 || t.ldd < t.nc || (t.src && t.lds < (t.transposed ? t.nr : t.nc)) || (!t.src && t.nr != t.nc))
            DMRGX_FAIL(DMRGX_ERR_ARG, "cells_axpy: task %d malformed", i);
        ht[i] = AxTask{t.dst, t.src, t.ldd, t.lds, t.nr, t.nc, t.transposed ? 1 : 0, 0, t.alpha};
        // conservative overlap rule: tasks sharing a destination base pointer are serialised
        rnd[i] = rounds[t.dst_base ? t.dst_base : t.dst]++;
        max_round = std::max(max_round, rnd[i]);
    }
    // the task table and the tile lists of every round go out in one copy
    DevBuf d_tab;
    PackedUpload pk;
    const size_t o_tasks = pk.add(ht);
    std::vector<std::pair<size_t, size_t>> lists;          // (offset, tiles) per non-empty round
    for (int32_t r = 0; r <= max_round; ++r) {
        std::vector<AxTile> tl;
        for (int32_t i = 0; i < n; ++i) if (rnd[i] == r)
            for (int32_t ti = 0; ti < (ht[i].nr + 31) / 32; ++ti) for (int32_t tj = 0; tj < (ht[i].nc + 31) / 32; ++tj) tl.push_back(AxTile{i, ti, tj, 0});
        if (tl.empty()) continue;
        lists.push_back({pk.add(tl), tl.size()});
    }
    DMRGX_CHK(pk.upload(d_tab, st));
    for (const auto& l : lists) {
        hipLaunchKernelGGL(cells_axpy_kernel, dim3((unsigned)l.second), dim3(256), 0, st, (const AxTile*)packed_at<AxTile>(d_tab, l.first), (const AxTask*)packed_at<AxTask>(d_tab, o_tasks));
        DMRGX_HIP(hipGetLastError());
    }
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_rotate_ops(const dmrgx_sectors* old_sectors, const dmrgx_rotation* rot, int32_t nops,
                                         const dmrgx_secop* src_ops, double* const* const* dst_blocks, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const auto h0 = std::chrono::steady_clock::now();
    if (!old_sectors || !rot || nops < 0 || (nops > 0 && (!src_ops || !dst_blocks))) DMRGX_FAIL(DMRGX_ERR_ARG, "rotate_ops: null argument");
    const int32_t nn = rot->n_new;
    if (nn <= 0 || !rot->old_sector || !rot->kept || !rot->rot_t) DMRGX_FAIL(DMRGX_ERR_ARG, "rotate_ops: empty rotation");
    std::vector<int32_t> new_of_old(old_sectors->nsec, -1);
    for (int32_t a = 0; a < nn; ++a) {
        const int32_t q = rot->old_sector[a];
        if (q < 0 || q >= old_sectors->nsec || (a > 0 && q <= rot->old_sector[a - 1])) DMRGX_FAIL(DMRGX_ERR_OUTOFRANGE, "rotate_ops: old_sector[%d]=%d not ascending/in range", a, q);
        if (rot->kept[a] <= 0 || rot->kept[a] > old_sectors->size[q] || !rot->rot_t[a]) DMRGX_FAIL(DMRGX_ERR_ARG, "rotate_ops: new sector %d keeps %d of %d", a, rot->kept[a], old_sectors->size[q]);
        new_of_old[q] = a;
    }
    // UT_a = RT_a^T  (n_q x m_a), one workspace
    std::vector<int64_t> ut_off(nn + 1, 0);
    for (int32_t a = 0; a < nn; ++a) ut_off[a + 1] = ut_off[a] + (int64_t)rot->kept[a] * old_sectors->size[rot->old_sector[a]];
    // W workspace
    struct WRef { int32_t op, a, ap, cell; int64_t off; };
    std::vector<WRef> wrefs;
    int64_t wtot = 0;
    for (int32_t o = 0; o < nops; ++o) {
        const dmrgx_secop& op = src_ops[o];
        if (op.transposed) DMRGX_FAIL(DMRGX_ERR_ARG, "rotate_ops: transposed source operators are not supported (rotate Sp, not Sm)");
        if (op.ncells < 0 || (op.ncells > 0 && !op.cells)) DMRGX_FAIL(DMRGX_ERR_ARG, "rotate_ops: operator %d has a bad cell list", o);
        for (int32_t c = 0; c < op.ncells; ++c) {
            const dmrgx_cell& ce = op.cells[c];
            const int32_t q = ce.row_sector, qc = q + op.shift;
            if (q < 0 || q >= old_sectors->nsec || qc < 0 || qc >= old_sectors->nsec) DMRGX_FAIL(DMRGX_ERR_OUTOFRANGE, "rotate_ops: op %d cell %d sector out of range", o, c);
            if (ce.nr <= 0 || ce.nc <= 0 || ce.r0 < 0 || ce.c0 < 0 || ce.r0 + ce.nr > old_sectors->size[q] || ce.c0 + ce.nc > old_sectors->size[qc])
                DMRGX_FAIL(DMRGX_ERR_OUTOFRANGE, "rotate_ops: op %d cell %d exceeds its block", o, c);
            if (ce.kind == DMRGX_CELL_DENSE && (!ce.data || ce.ld < ce.nc)) DMRGX_FAIL(DMRGX_ERR_ARG, "rotate_ops: op %d cell %d has no data", o, c);
            const int32_t a = new_of_old[q], ap = new_of_old[qc];
            if (a < 0 || ap < 0) continue;                      // a truncated-away sector: the block disappears
            if (!dst_blocks[o] || !dst_blocks[o][a]) DMRGX_FAIL(DMRGX_ERR_ARG, "rotate_ops: op %d has no destination for new sector %d", o, a);
            wrefs.push_back(WRef{o, a, ap, c, wtot});
            wtot += (int64_t)rot->kept[a] * ce.nc;
        }
    }
    DevBuf ws;
    DMRGX_CHK(ws.alloc((size_t)std::max<int64_t>(ut_off[nn] + wtot, 1) * sizeof(double)));
    double* UT = ws.as<double>();
    double* W = UT + ut_off[nn];
    {
        std::vector<dmrgx_axpy_task> tr(nn);
        DMRGX_HIP(zero_async(UT, (size_t)ut_off[nn] * sizeof(double), st));
        for (int32_t a = 0; a < nn; ++a) {
            const int32_t nq = old_sectors->size[rot->old_sector[a]], m = rot->kept[a];
            tr[a] = dmrgx_axpy_task{UT + ut_off[a], nullptr, rot->rot_t[a], m, nq, nq, m, 1, 1.0};
        }
        DMRGX_CHK(dmrgx_cells_axpy(nn, tr.data(), st));
    }
    std::vector<GProd> prods;
    std::vector<GGroup> groups;
    std::vector<GTile> tA, tAb, tB, tBb;
    // stage A
    for (const WRef& w : wrefs) {
        const dmrgx_cell& ce = src_ops[w.op].cells[w.cell];
        const int32_t nq = old_sectors->size[rot->old_sector[w.a]], m = rot->kept[w.a];
        const double* RTa = rot->rot_t[w.a] + ce.r0;
        const int32_t pb = (int32_t)prods.size();
        if (ce.kind == DMRGX_CELL_DENSE) prods.push_back(GProd{RTa, ce.data, nq, (int32_t)ce.ld, ce.nr, GPROD_GEMM, 1.0});
        else prods.push_back(GProd{nullptr, RTa, 0, nq, 0, GPROD_AXPY, ce.scale});
        groups.push_back(GGroup{W + w.off, ce.nc, m, ce.nc, pb, pb + 1, ce.kind == DMRGX_CELL_DENSE ? 0 : 1, 0});
        ggemm_append_tiles_mixed(tAb, tA, (int32_t)groups.size() - 1, m, ce.nc, (ce.nr + GG_BK - 1) / GG_BK);
    }
    // stage B: group per (op, a) destination
    std::map<std::pair<int32_t, int32_t>, std::vector<const WRef*>> by_dst;
    for (const WRef& w : wrefs) by_dst[{w.op, w.a}].push_back(&w);
    // every destination block that exists is written, also when no source cell reaches it (then it is zero)
    for (int32_t o = 0; o < nops; ++o)
        for (int32_t a = 0; a < nn; ++a) {
            const int32_t qc = rot->old_sector[a] + src_ops[o].shift;
            if (qc < 0 || qc >= old_sectors->nsec || new_of_old[qc] < 0 || !dst_blocks[o] || !dst_blocks[o][a]) continue;
            by_dst[{o, a}];
        }
    for (auto& kv : by_dst) {
        const int32_t o = kv.first.first, a = kv.first.second, ap = new_of_old[rot->old_sector[a] + src_ops[o].shift];
        const int32_t m = rot->kept[a], mp = rot->kept[ap];
        const int32_t pb = (int32_t)prods.size();
        int32_t cost = 1;
        for (const WRef* w : kv.second) {
            const dmrgx_cell& ce = src_ops[o].cells[w->cell];
            prods.push_back(GProd{W + w->off, UT + ut_off[ap] + (int64_t)ce.c0 * mp, ce.nc, mp, ce.nc, GPROD_GEMM, 1.0});
            cost += (ce.nc + GG_BK - 1) / GG_BK;
        }
        groups.push_back(GGroup{dst_blocks[o][a], mp, m, mp, pb, (int32_t)prods.size(), 0, 0});
        ggemm_append_tiles_mixed(tBb, tB, (int32_t)groups.size() - 1, m, mp, cost);
    }
    const auto h1 = std::chrono::steady_clock::now();
    // stage A goes to the device before stage B's tile lists are scheduled: the host work of B (a sort over ~10^4 tiles) then
    // runs behind A's GEMMs instead of in front of an idle GPU
    ggemm_schedule(tA, groups); ggemm_schedule(tAb, groups, 2);
    const auto h2 = std::chrono::steady_clock::now();
    DevBuf dtabA, dtabB;
    PackedUpload pkA;
    const size_t o_p = pkA.add(prods), o_g = pkA.add(groups), o_1 = pkA.add(tAb), o_2 = pkA.add(tA);
    DMRGX_CHK(pkA.upload(dtabA, st));
    const GProd* dp = packed_at<GProd>(dtabA, o_p);
    const GGroup* dg = packed_at<GGroup>(dtabA, o_g);
    const auto h3 = std::chrono::steady_clock::now();
    static const bool trace = getenv("DMRGX_ROT_TRACE") != nullptr;      // developer aid: flops and time of the two stages
    if (trace) fprintf(stderr, "[rotate] host: tables %.3f ms, schedule A %.3f ms, uploads A %.3f ms\n", std::chrono::duration<double, std::milli>(h1 - h0).count(),
                       std::chrono::duration<double, std::milli>(h2 - h1).count(), std::chrono::duration<double, std::milli>(h3 - h2).count());
    hipEvent_t ev[3];
    if (trace) { for (auto& e : ev) DMRGX_HIP(hipEventCreate(&e)); DMRGX_HIP(hipEventRecord(ev[0], st)); }
    DMRGX_CHK(ggemm_launch(packed_at<GTile>(dtabA, o_1), dg, dp, (int32_t)tAb.size(), st, 1));
    DMRGX_CHK(ggemm_launch(packed_at<GTile>(dtabA, o_2), dg, dp, (int32_t)tA.size(), st, 0));
    if (trace) DMRGX_HIP(hipEventRecord(ev[1], st));
    ggemm_schedule(tB, groups); ggemm_schedule(tBb, groups, 2);
    PackedUpload pkB;
    const size_t o_3 = pkB.add(tBb), o_4 = pkB.add(tB);
    DMRGX_CHK(pkB.upload(dtabB, st));
    DMRGX_CHK(ggemm_launch(packed_at<GTile>(dtabB, o_3), dg, dp, (int32_t)tBb.size(), st, 1));
    DMRGX_CHK(ggemm_launch(packed_at<GTile>(dtabB, o_4), dg, dp, (int32_t)tB.size(), st, 0));
    if (trace) {
        DMRGX_HIP(hipEventRecord(ev[2], st));
        DMRGX_HIP(hipEventSynchronize(ev[2]));
        double fa = 0, fb = 0;
        for (const GGroup& g : groups) for (int32_t q = g.prod_begin + g.n_axpy; q < g.prod_end; ++q) ((&g - groups.data()) < (ptrdiff_t)wrefs.size() ? fa : fb) += 2.0 * g.M * g.N * prods[q].K;
        float ma = 0, mb = 0;
        DMRGX_HIP(hipEventElapsedTime(&ma, ev[0], ev[1])); DMRGX_HIP(hipEventElapsedTime(&mb, ev[1], ev[2]));
        fprintf(stderr, "[rotate] %d ops: stage A %.2f GF %.3f ms (%.1f TF/s, tiles %zu+%zu big), stage B %.2f GF %.3f ms (%.1f TF/s, tiles %zu+%zu big)\n", nops,
                fa * 1e-9, ma, fa / (ma * 1e9), tA.size(), tAb.size(), fb * 1e-9, mb, fb / (mb * 1e9), tB.size(), tBb.size());
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    return DMRGX_OK;
}
