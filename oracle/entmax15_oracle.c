/* ORACLE (test infrastructure, NOT product code): CPU restatement of 1.5-entmax.
 *
 * The reference calls the third-party package `entmax` (entmax15 at lvmhelpers/marg.py:46, the
 * default normaliser of ExplicitWrapper; Entmax15Loss at
 * experiments/semi_supervised-vae/train.py:9,314-315).  The package is not in /root/reference and
 * not importable here, so this file restates its PUBLISHED algorithm (Peters, Niculae, Martins,
 * "Sparse Sequence-to-Sequence Models", ACL 2019, Algorithm 2; entmax 1.0
 * `Entmax15Function._threshold_and_support`): PARITY UNPINNED by the reference -- pinned instead by
 * the torch restatement in tests/test_oracle_entmax15.py, by sum p = 1 and by the optimality
 * conditions p_i = max(x_i/2 - tau, 0)^2.
 *
 *   z = x/2 - max(x/2);  sort z descending;  for every prefix j (1-based):
 *     mean_j = S1_j / j,  ss_j = j (S2_j/j - mean_j^2),  delta_j = (1 - ss_j)/j,
 *     tau_j = mean_j - sqrt(max(delta_j, 0));
 *   support = #{ j : tau_j <= z_(j) },  tau* = tau_support,  p = max(z - tau*, 0)^2.
 * Prefix sums in fp64 (the package runs torch.cumsum in the input dtype; tolerance tests, not
 * bit-exact ones, compare against this file).
 *
 * Backward (Entmax15Function.backward): g = sqrt(p); dx = dy g - (sum dy g / sum g) g.
 * Loss (Entmax15Loss, reduction 'none'): (1 - sum p sqrt(p)) / 0.75 + <p - e_y, x>, gradient p - e_y.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static int cmp_desc(const void* a, const void* b)
{
    const float x = *(const float*)a, y = *(const float*)b;
    return (x < y) - (x > y);
}

int oracle_entmax15_fwd(const float* x, float* p, int32_t* supp, float* tau_out, int64_t R, int32_t K)
{
    float* z = (float*)malloc(sizeof(float) * (size_t)K);
    float* s = (float*)malloc(sizeof(float) * (size_t)K);
    if (!z || !s) return -1;
    for (int64_t r = 0; r < R; ++r) {
        const float* xr = x + r * K;
        float mx = xr[0] * 0.5f;
        for (int j = 1; j < K; ++j) mx = fmaxf(mx, xr[j] * 0.5f);
        for (int j = 0; j < K; ++j) s[j] = z[j] = xr[j] * 0.5f - mx;
        qsort(s, (size_t)K, sizeof(float), cmp_desc);
        double S1 = 0.0, S2 = 0.0, tau_star = 0.0;
        int support = 0;
        for (int j = 1; j <= K; ++j) {
            const double v = (double)s[j - 1];
            S1 += v;
            S2 += v * v;
            const double mean = S1 / j, ss = j * (S2 / j - mean * mean), delta = (1.0 - ss) / j;
            const double t = mean - sqrt(delta > 0.0 ? delta : 0.0);
            if (t <= v) { ++support; }
        }
        /* tau at the support size (the package gathers tau[support - 1]) */
        S1 = S2 = 0.0;
        for (int j = 1; j <= support; ++j) {
            const double v = (double)s[j - 1];
            S1 += v;
            S2 += v * v;
        }
        {
            const double mean = S1 / support, ss = support * (S2 / support - mean * mean);
            const double delta = (1.0 - ss) / support;
            tau_star = mean - sqrt(delta > 0.0 ? delta : 0.0);
        }
        const float tf = (float)tau_star;
        int nz = 0;
        for (int j = 0; j < K; ++j) {
            const float d = z[j] - tf;
            const float q = d > 0.0f ? d * d : 0.0f;
            p[r * K + j] = q;
            nz += q > 0.0f;
        }
        if (supp) supp[r] = nz;
        if (tau_out) tau_out[r] = tf;
    }
    free(z);
    free(s);
    return 0;
}

int oracle_entmax15_bwd(const float* dy, const float* p, float* dx, int64_t R, int32_t K)
{
    for (int64_t r = 0; r < R; ++r) {
        double a = 0.0, b = 0.0;
        for (int j = 0; j < K; ++j) {
            const double g = sqrt((double)p[r * K + j]);
            a += (double)dy[r * K + j] * g;
            b += g;
        }
        const double q = a / b;
        for (int j = 0; j < K; ++j) {
            const double g = sqrt((double)p[r * K + j]);
            dx[r * K + j] = (float)(((double)dy[r * K + j] - q) * g);
        }
    }
    return 0;
}

int oracle_entmax15_loss(const float* x, const int64_t* target, float* loss, float* grad, int64_t R,
                         int32_t K)
{
    if (oracle_entmax15_fwd(x, grad, 0, 0, R, K)) return -1;
    for (int64_t r = 0; r < R; ++r) {
        double omega = 0.0, dot = 0.0;
        for (int j = 0; j < K; ++j) {
            const double q = (double)grad[r * K + j];
            omega += q * sqrt(q);
        }
        grad[r * K + target[r]] -= 1.0f;
        for (int j = 0; j < K; ++j) dot += (double)grad[r * K + j] * (double)x[r * K + j];
        loss[r] = (float)((1.0 - omega) / 0.75 + dot);
    }
    return 0;
}
