/* Oracle (TEST INFRASTRUCTURE ONLY): C restatement of the reference's matrix-free superblock MatMult.
 *
 *   oracle_kron_apply_ref  == the row loop of MatMult_KronSumShell, src/DMRGKron.cpp:1842-1864:
 *       y[ir] = sum_it a[it] * sum_l vL[l] * sum_r vR[r] * x[(idxL[l]-bks_L)*col_NStatesR + fws_O + idxR[r]]
 *   with the per-(row,term) descriptors of KronSumSetUpShellTerms (src/DMRGKron.cpp:1706-1824) stored as
 *   per-(KronBlock,term) constants (they only change when the iterator enters a new KronBlock, :1751-1767)
 *   plus the CSR rows of the term's operators.  Identity factors are 1-entry rows (:1781-1797).
 *   Rows are split over OpenMP threads in contiguous, predicted-nnz-balanced ranges, the shared-memory
 *   equivalent of KronSumShellSplitOwnership (src/DMRGKron.cpp:1519-1704).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC oracle/kron_ref.c -o oracle/liboracle_kron.so
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int64_t N;              /* superblock dimension (rows of the target sector)            */
    int32_t nterms;         /* 2 + number of LR terms                                       */
    int32_t nblocks;        /* number of KronBlocks                                         */
    const double *term_a;   /* [nterms]                                                     */
    const int64_t *const *A_indptr, *const *A_indices; const double *const *A_data;   /* per term */
    const int64_t *const *B_indptr, *const *B_indices; const double *const *B_data;   /* per term */
    const int64_t *Rows_L, *Rows_R;      /* [N] global row of the left/right operator       */
    const int64_t *blk_of_row;           /* [N] KronBlock index of the row                  */
    const int64_t *bks_L, *col_NStatesR, *fws_O, *valid;  /* [nblocks*nterms]              */
} oracle_shell_ctx;

/* predicted cost of one row = sum_t nz_L*nz_R (the reference's ks_nnz, src/DMRGKron.cpp:1560-1640) */
static int64_t row_cost(const oracle_shell_ctx *c, int64_t ir)
{
    const int64_t k = c->blk_of_row[ir], rl = c->Rows_L[ir], rr = c->Rows_R[ir];
    int64_t cost = 1;
    for (int32_t it = 0; it < c->nterms; ++it) {
        if (!c->valid[k * c->nterms + it]) continue;
        cost += (c->A_indptr[it][rl + 1] - c->A_indptr[it][rl]) * (c->B_indptr[it][rr + 1] - c->B_indptr[it][rr]);
    }
    return cost;
}

static void apply_rows(const oracle_shell_ctx *c, const double *x, double *y, int64_t r0, int64_t r1)
{
    for (int64_t ir = r0; ir < r1; ++ir) {
        const int64_t k = c->blk_of_row[ir], rl = c->Rows_L[ir], rr = c->Rows_R[ir];
        double yval = 0.0;
        for (int32_t it = 0; it < c->nterms; ++it) {
            const int64_t kt = k * c->nterms + it;
            if (!c->valid[kt]) continue;                       /* nz_L = nz_R = 0  (:1799-1801) */
            const int64_t *idxL = c->A_indices[it]; const double *vL = c->A_data[it];
            const int64_t *idxR = c->B_indices[it]; const double *vR = c->B_data[it];
            const int64_t l0 = c->A_indptr[it][rl], l1 = c->A_indptr[it][rl + 1];
            const int64_t q0 = c->B_indptr[it][rr], q1 = c->B_indptr[it][rr + 1];
            const int64_t bks = c->bks_L[kt], ncr = c->col_NStatesR[kt], fws = c->fws_O[kt];
            const double a = c->term_a[it];
            for (int64_t l = l0; l < l1; ++l) {
                const int64_t idx = (idxL[l] - bks) * ncr + fws;      /* :1855 */
                const double temp = a * vL[l];                         /* :1856 */
                for (int64_t r = q0; r < q1; ++r)
                    yval += temp * vR[r] * x[idx + idxR[r]];           /* :1859 */
            }
        }
        y[ir] = yval;                                                  /* :1863 */
    }
}

/* y[row_begin:row_end] = (H x)[row_begin:row_end]; nthreads<=0 -> OpenMP default. Returns threads used. */
int oracle_kron_apply_ref(const oracle_shell_ctx *c, const double *x, double *y,
                          int64_t row_begin, int64_t row_end, int nthreads)
{
    int nt = 1;
#ifdef _OPENMP
    nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#endif
    if (row_end - row_begin < nt) nt = 1;
    /* contiguous cost-balanced ranges */
    int64_t *cut = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nt + 1));
    if (nt == 1) { cut[0] = row_begin; cut[1] = row_end; }
    else {
        int64_t tot = 0;
        for (int64_t ir = row_begin; ir < row_end; ++ir) tot += row_cost(c, ir);
        int64_t acc = 0; int p = 1; cut[0] = row_begin;
        for (int64_t ir = row_begin; ir < row_end && p < nt; ++ir) {
            acc += row_cost(c, ir);
            while (p < nt && acc * nt >= tot * p) cut[p++] = ir + 1;
        }
        while (p <= nt) cut[p++] = row_end;
        cut[nt] = row_end;
    }
#ifdef _OPENMP
#pragma omp parallel for num_threads(nt) schedule(static, 1)
#endif
    for (int p = 0; p < nt; ++p) apply_rows(c, x, y, cut[p], cut[p + 1]);
    free(cut);
    return nt;
}

/* unfactored flop count of the literal loop (2 flops per innermost update + 1 per l), for reporting */
int64_t oracle_kron_ref_flops(const oracle_shell_ctx *c, int64_t row_begin, int64_t row_end)
{
    int64_t f = 0;
    for (int64_t ir = row_begin; ir < row_end; ++ir) f += 2 * (row_cost(c, ir) - 1);
    return f;
}
