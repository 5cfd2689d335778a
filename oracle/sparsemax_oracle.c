/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of the sparsemax arithmetic used on the reference hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 *
 * What it restates
 * ----------------
 * The reference calls the third-party package `entmax` (unpinned, requirements.txt:2), which is
 * NOT in /root/reference and not installed here:
 *     lvmhelpers/marg.py:3,45,51          sparsemax(scores, dim=-1)
 *     lvmhelpers/structmarg.py:3,48       sparsemax(scores, dim=-1)
 *     experiments/semi_supervised-vae/train.py:9,315-317   SparsemaxLoss(reduction='none')
 * so this file restates entmax's PUBLISHED sort-based algorithm (Martins & Astudillo 2016,
 * Alg. 1; entmax 1.0 `SparsemaxFunction`), with torch-CPU rounding semantics (config 1 of
 * BASELINE.json is "reference CPU path"):
 *     z      = x - max(x)                                   (fp32)
 *     zs     = sort(z, descending)
 *     cs_j   = (float) sum_{i<=j} zs_i  accumulated in fp64 (torch CPU cumsum acc type), j=1..K
 *     c_j    = cs_j - 1.0f                                  (fp32)
 *     s      = #{ j : (float)j * zs_j > c_j }               (fp32 multiply, strict compare)
 *     tau    = c_s / (float)s                               (fp32)
 *     p      = max(z - tau, 0)                              (fp32)
 * backward (support-masked mean-centred JVP):
 *     g = dy * [p != 0];  v = sum(g) / (float)s;  dx = [p != 0] * (g - v)
 *
 * PARITY UNPINNED: the reference holds no tests/golden vectors for this arithmetic and entmax is
 * absent, so this restatement is pinned only by mathematical invariants (tests/test_oracle_*.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

static int cmp_desc(const void* a, const void* b)
{
    float fa = *(const float*)a, fb = *(const float*)b;
    return (fa < fb) - (fa > fb);
}

/* x[R,K] -> p[R,K], supp[R] (support_size s), tau[R] (tau relative to the max-shifted row;
 * nullable), nnz[R] (#{p > 0}; nullable) */
int oracle_sparsemax_fwd(const float* x, float* p, int32_t* supp, float* tau_out, int32_t* nnz,
                         int64_t R, int32_t K)
{
    float* zs = (float*)malloc(sizeof(float) * (size_t)K);
    if (!zs) return -1;
    for (int64_t r = 0; r < R; ++r) {
        const float* xr = x + r * (int64_t)K;
        float* pr = p + r * (int64_t)K;
        float mx = xr[0];
        for (int32_t j = 1; j < K; ++j) if (xr[j] > mx) mx = xr[j];
        for (int32_t j = 0; j < K; ++j) { pr[j] = xr[j] - mx; zs[j] = pr[j]; }
        qsort(zs, (size_t)K, sizeof(float), cmp_desc);
        double acc = 0.0;
        int32_t s = 0;
        float c_at_count = 0.0f;
        /* two passes are folded: count the TRUE conditions (entmax sums the boolean mask, it
         * does not stop at the first false one), then tau = c[s-1]/s. */
        float* c = zs; /* reuse storage: c_j overwrites zs_j after use */
        for (int32_t j = 0; j < K; ++j) {
            acc += (double)zs[j];
            float cj = (float)acc - 1.0f;
            float lhs = (float)(j + 1) * zs[j];
            if (lhs > cj) ++s;
            c[j] = cj;
        }
        if (s < 1) s = 1; /* cannot happen for finite input: j=1 gives 0 > -1 */
        c_at_count = c[s - 1];
        float tau = c_at_count / (float)s;
        int32_t cnt = 0;
        for (int32_t j = 0; j < K; ++j) {
            float v = pr[j] - tau;
            v = v > 0.0f ? v : 0.0f;
            pr[j] = v;
            cnt += (v > 0.0f);
        }
        supp[r] = s;
        if (tau_out) tau_out[r] = tau;
        if (nnz) nnz[r] = cnt;
    }
    free(zs);
    return 0;
}

/* dy[R,K], p[R,K], supp[R] -> dx[R,K] */
int oracle_sparsemax_bwd(const float* dy, const float* p, const int32_t* supp, float* dx,
                         int64_t R, int32_t K)
{
    for (int64_t r = 0; r < R; ++r) {
        const float* g = dy + r * (int64_t)K;
        const float* pr = p + r * (int64_t)K;
        float* o = dx + r * (int64_t)K;
        double acc = 0.0; /* fp64 accumulate, rounded once: at least as accurate as torch's fp32 sum */
        for (int32_t j = 0; j < K; ++j) if (pr[j] != 0.0f) acc += (double)g[j];
        float v = (float)acc / (float)supp[r];
        for (int32_t j = 0; j < K; ++j) o[j] = (pr[j] != 0.0f) ? (g[j] - v) : 0.0f;
    }
    return 0;
}

/* lvmhelpers/marg.py:6-28: -sum_{p>0} pc*log(pc), pc = clamp(p, eps, 1-eps), eps = FLT_EPSILON */
int oracle_entropy(const float* p, float* ent, int64_t R, int32_t K)
{
    const float eps = FLT_EPSILON;
    for (int64_t r = 0; r < R; ++r) {
        const float* pr = p + r * (int64_t)K;
        double acc = 0.0;
        for (int32_t j = 0; j < K; ++j) {
            if (pr[j] > 0.0f) {
                float pc = pr[j];
                if (pc < eps) pc = eps;
                if (pc > 1.0f - eps) pc = 1.0f - eps;
                acc += (double)(pc * logf(pc));
            }
        }
        ent[r] = -(float)acc;
    }
    return 0;
}

/* Fenchel-Young sparsemax loss (entmax SparsemaxLoss, reduction='none'):
 *   L(z, y) = (1 - ||p||^2)/2 + <p, z> - z_y   restated from the published formula
 *   (computed on the max-shifted row like entmax: z := x - max is NOT applied by entmax's loss,
 *   it uses raw X; the value is shift-invariant mathematically). Gradient: p - e_y. */
int oracle_sparsemax_loss(const float* x, const int64_t* target, float* loss, float* grad,
                          int64_t R, int32_t K)
{
    float* p = (float*)malloc(sizeof(float) * (size_t)K);
    int32_t s;
    if (!p) return -1;
    for (int64_t r = 0; r < R; ++r) {
        const float* xr = x + r * (int64_t)K;
        oracle_sparsemax_fwd(xr, p, &s, NULL, NULL, 1, K);
        double acc = 0.0;
        for (int32_t j = 0; j < K; ++j)
            if (p[j] > 0.0f) acc += (double)p[j] * ((double)xr[j] - 0.5 * (double)p[j]);
        acc += 0.5 - (double)xr[target[r]];
        loss[r] = (float)acc;
        if (grad) {
            float* gr = grad + r * (int64_t)K;
            for (int32_t j = 0; j < K; ++j) gr[j] = p[j];
            gr[target[r]] -= 1.0f;
        }
    }
    free(p);
    return 0;
}

/* Row-major support compaction, identical to torch.nonzero(p > 0) / boolean-mask indexing
 * (lvmhelpers/structmarg.py:113-126). Returns N. rows/cols/vals may be NULL to only count. */
int64_t oracle_support_compact(const float* p, int64_t* row_offsets, int32_t* rows, int32_t* cols,
                               float* vals, int64_t R, int32_t K)
{
    int64_t n = 0;
    for (int64_t r = 0; r < R; ++r) {
        if (row_offsets) row_offsets[r] = n;
        for (int32_t j = 0; j < K; ++j) {
            float v = p[r * (int64_t)K + j];
            if (v > 0.0f) {
                if (rows) rows[n] = (int32_t)r;
                if (cols) cols[n] = j;
                if (vals) vals[n] = v;
                ++n;
            }
        }
    }
    if (row_offsets) row_offsets[R] = n;
    return n;
}

/* out[r] = sum_{j in row r} vals[j]*comps[j] (fp64 accumulate); returns sum_r out[r] * inv_b
 * (lvmhelpers/marg.py:125-126, structmarg.py:140,242). */
double oracle_weighted_loss_reduce(const float* vals, const float* comps, const int64_t* row_offsets,
                                   float* out, int64_t R, double inv_b)
{
    double tot = 0.0;
    for (int64_t r = 0; r < R; ++r) {
        double acc = 0.0;
        for (int64_t j = row_offsets[r]; j < row_offsets[r + 1]; ++j)
            acc += (double)vals[j] * (double)comps[j];
        if (out) out[r] = (float)acc;
        tot += acc;
    }
    return tot * inv_b;
}
