// Kernel 3: bit-vector SparseMAP -- active-set QP solver, one wavefront (= one workgroup of 64
// lanes) per sample, everything the solver touches resident in LDS.
//
// Replaces, for a whole batch, lvmhelpers/sparsemap.py:12-36 (PFactorGraph construction,
// init_active_set_from_scores, lpsmap GenericFactor::SolveQP, get_sparse_solution) with the linear
// oracles of lvmhelpers/bernoulli.h:14-35 (Bernoulli) and lvmhelpers/budget.h:24-85 (budget), and
// lvmhelpers/sparsemap.py:38-44 (dist_jacobian_vec) for the backward.
//
// The arithmetic follows oracle/sparsemap_oracle.c choice by choice (C1..C6 there) so that the
// active set -- which is path dependent -- comes out identical: every fp64 inner product that
// feeds a decision is accumulated serially in the oracle's order by ONE lane (lane c owns output
// c; sums that produce a single scalar are accumulated redundantly by all lanes from LDS
// broadcasts), there is no FMA contraction (this file is built with -ffp-contract=off) and
// division is IEEE.  Parallelism comes from the 64 lanes owning different outputs, from the
// rank-one updates of the cached inverse KKT matrix, and from one workgroup per sample.
//
// LDS per sample (n = bits, cap = n+1 configurations, ld = n+2): the (ld x ld) fp64 inverse KKT
// matrix dominates: 9.2 KB at n = 32, 135 KB at n = 128 (one sample per CU at a time).
#include "common.hpp"

#include <math.h>

namespace smlvm {

enum { SM_CONVERGED = 0, SM_MAXITER = 1, SM_SINGULAR = 2, SM_DUPLICATE = 3 };

struct SmLayout {
    int n, cap, ld, W;
    // offsets in bytes
    size_t A, z, dist, bsc, d, r, on, off, term, e, eta, sorted, cfg, newc, total;
    __host__ __device__ SmLayout(int n_) : n(n_), cap(n_ + 1), ld(n_ + 2), W((n_ + 31) / 32)
    {
        size_t o = 0;
        A = o;      o += sizeof(double) * (size_t)ld * ld;
        z = o;      o += sizeof(double) * ld;
        dist = o;   o += sizeof(double) * ld;
        bsc = o;    o += sizeof(double) * ld;
        d = o;      o += sizeof(double) * ld;
        r = o;      o += sizeof(double) * ld;
        on = o;     o += sizeof(double) * n;
        off = o;    o += sizeof(double) * n;
        term = o;   o += sizeof(double) * n;
        e = o;      o += sizeof(double) * n;
        eta = o;    o += sizeof(double) * n;
        sorted = o; o += sizeof(double) * n;
        cfg = o;    o += sizeof(uint32_t) * (size_t)cap * W;
        newc = o;   o += sizeof(uint32_t) * ((W + 1) & ~1);
        total = (o + 15) & ~(size_t)15;
    }
};

__device__ __forceinline__ void wsync() { __syncthreads(); }  // single-wave workgroup

// Diagnostic build only (tools/sm_prof.hip defines SMLVM_SM_PROF): cycles per solver phase.
#ifdef SMLVM_SM_PROF
__device__ unsigned long long g_prof[16];
#define SM_T(k)                                                                   \
    do {                                                                          \
        long long now_ = clock64();                                               \
        if (threadIdx.x == 0) atomicAdd(&g_prof[k], (unsigned long long)(now_ - t_last_)); \
        t_last_ = now_;                                                           \
    } while (0)
#define SM_T0() long long t_last_ = clock64()
#else
#define SM_T(k) do { } while (0)
#define SM_T0() do { } while (0)
#endif

__device__ __forceinline__ double bcast_f64(double v, int lane)
{
    long long b = __builtin_bit_cast(long long, v);
    int lo = __shfl((int)(b & 0xffffffffll), lane);
    int hi = __shfl((int)(b >> 32), lane);
    b = ((long long)hi << 32) | (unsigned int)lo;
    return __builtin_bit_cast(double, b);
}

// serial left-to-right sum of an LDS array, same bits in every lane
__device__ __forceinline__ double serial_sum(const double* a, int len)
{
    double v = 0.0;
#pragma unroll 16
    for (int k = 0; k < len; ++k) v += a[k];
    return v;
}

// Linear oracle on potentials on[]/off[] (LDS) -> newc[] (LDS bit words), returns the value.
// budget <= 0: lvmhelpers/bernoulli.h:14-35; budget > 0: lvmhelpers/budget.h:24-85.
__device__ double maximize(const SmLayout& L, char* sm, int budget, int lane)
{
    const double* on = (const double*)(sm + L.on);
    const double* off = (const double*)(sm + L.off);
    double* term = (double*)(sm + L.term);
    double* e = (double*)(sm + L.e);
    double* sorted = (double*)(sm + L.sorted);
    uint32_t* newc = (uint32_t*)(sm + L.newc);
    const int n = L.n;
    double value;
    if (budget <= 0) {
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            bool bit = false;
            if (k < n) {
                const double a = on[k], b = off[k];
                bit = a > b;
                term[k] = bit ? a : b;
            }
            const unsigned long long m = __ballot(bit);
            if (lane == 0) {
                newc[(k0 >> 5)] = (uint32_t)m;
                if ((k0 >> 5) + 1 < L.W) newc[(k0 >> 5) + 1] = (uint32_t)(m >> 32);
            }
        }
        wsync();
        value = serial_sum(term, n);
        wsync();
        return value;
    }
    // budget factor
    int num_active = 0;
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        bool pos = false;
        if (k < n) {
            const double ev = on[k] - off[k];
            e[k] = ev;
            pos = ev > 0;
            term[k] = pos ? ev : 0.0;
        }
        const unsigned long long m = __ballot(pos);
        num_active += __popcll(m);
        if (lane == 0) {
            newc[(k0 >> 5)] = (uint32_t)m;
            if ((k0 >> 5) + 1 < L.W) newc[(k0 >> 5) + 1] = (uint32_t)(m >> 32);
        }
    }
    wsync();
    value = 0.0;
    double sum = 0.0;  // adding 0.0 for the inactive ones changes nothing
#pragma unroll 8
    for (int k = 0; k < n; ++k) {  // two independent serial chains, interleaved
        value += off[k];
        sum += term[k];
    }
    if (num_active > budget) {
        // rank in the order of sort((-e_i, i)) ascending: e descending, ties by smaller index
        int cnt_sel = 0;
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            bool sel = false;
            if (k < n) {
                const double ek = e[k];
                int rank = 0;
#pragma unroll 8
                for (int j = 0; j < n; ++j) {
                    const double ej = e[j];
                    rank += ((-ej < -ek) || (!(-ek < -ej) && j < k)) ? 1 : 0;
                }
                sorted[rank] = ek;
                sel = (rank < budget) && !(ek < 0);
            }
            const unsigned long long m = __ballot(sel);
            cnt_sel += __popcll(m);
            if (lane == 0) {
                newc[(k0 >> 5)] = (uint32_t)m;
                if ((k0 >> 5) + 1 < L.W) newc[(k0 >> 5) + 1] = (uint32_t)(m >> 32);
            }
        }
        wsync();
        sum = serial_sum(sorted, cnt_sel);
    }
    wsync();
    return value + sum;
}

// Evaluate (bernoulli.h:38-54) under eta = [x; 0]: sum over the on-bits in index order.
__device__ __forceinline__ double evaluate(const SmLayout& L, char* sm, const uint32_t* c)
{
    const double* eta = (const double*)(sm + L.eta);
    double v = 0.0;
    for (int w = 0; w < L.W; ++w) {
        const uint32_t bits = c[w];
        const int k0 = w * 32, kn = (L.n - k0 < 32) ? (L.n - k0) : 32;
#pragma unroll 8
        for (int b = 0; b < kn; ++b) v += ((bits >> b) & 1u) ? eta[k0 + b] : 0.0;
    }
    return v;
}

__global__ __launch_bounds__(64) void sparsemap_fwd_kernel(
    const float* __restrict__ scores, const double* __restrict__ init_scores, int budget,
    int max_iter, int64_t B, int n, uint32_t* __restrict__ aset_bits, float* __restrict__ p_out,
    int32_t* __restrict__ alen, double* __restrict__ inv_a, int32_t* __restrict__ iters,
    int32_t* __restrict__ status)
{
    extern __shared__ __align__(16) char sm[];
    const SmLayout L(n);
    const int lane = threadIdx.x;
    const int64_t s = blockIdx.x;
    const int cap = L.cap, ld = L.ld, W = L.W;
    double* A = (double*)(sm + L.A);
    double* z = (double*)(sm + L.z);
    double* dist = (double*)(sm + L.dist);
    double* bsc = (double*)(sm + L.bsc);
    double* d = (double*)(sm + L.d);
    double* r = (double*)(sm + L.r);
    double* on = (double*)(sm + L.on);
    double* off = (double*)(sm + L.off);
    double* eta = (double*)(sm + L.eta);
    uint32_t* cfg = (uint32_t*)(sm + L.cfg);
    uint32_t* newc = (uint32_t*)(sm + L.newc);

    for (int k = lane; k < n; k += 64) {
        const double x = (double)scores[s * n + k];
        eta[k] = x;
        if (init_scores) {  // InitActiveSetFromScores: Maximize(init)
            on[k] = init_scores[s * 2 * n + k];
            off[k] = init_scores[s * 2 * n + n + k];
        } else {            // SolveQP on an empty active set: Maximize(eta)
            on[k] = x;
            off[k] = 0.0;
        }
    }
    wsync();
    (void)maximize(L, sm, budget, lane);
    for (int w = lane; w < W; w += 64) cfg[w] = newc[w];
    if (lane == 0) {
        dist[0] = 1.0;
        A[0 * ld + 0] = -(double)n; A[0 * ld + 1] = 1.0; A[1 * ld + 0] = 1.0; A[1 * ld + 1] = 0.0;
    }
    wsync();
    {
        const double v = evaluate(L, sm, cfg);
        if (lane == 0) { bsc[0] = 1.0; bsc[1] = v; }
    }
    int m = 1;
    wsync();

    SM_T0();
    bool changed = true;
    int st = SM_MAXITER, it;
    double tau = 0.0;
    for (it = 0; it < max_iter; ++it) {
        bool same = true;
        if (changed) {
            // [tau; z] = invA^T b, b = [1; Evaluate(cfg_i)]
            double t0 = 0.0;
            for (int c = lane; c <= m; c += 64) {
                double acc = 0.0;
#pragma unroll 8
                for (int j = 0; j <= m; ++j) acc += A[j * ld + c] * bsc[j];
                if (c == 0) t0 = acc; else z[c - 1] = acc;
            }
            tau = bcast_f64(t0, 0);
            same = false;
            wsync();
            SM_T(0);
        }
        if (same) {
            for (int k = lane; k < n; k += 64) {
                double mon = 0.0, moff = 0.0;
                const uint32_t* cw = cfg + (k >> 5);
                const int sh = k & 31;
#pragma unroll 8
                for (int i = 0; i < m; ++i) {
                    const double zi = z[i];
                    const bool bit = (cw[i * W] >> sh) & 1u;
                    // x + 0.0 == x bit for bit (the sums never hold -0.0): add to both chains
                    mon += bit ? zi : 0.0;
                    moff += bit ? 0.0 : zi;
                }
                on[k] = eta[k] - mon;
                off[k] = 0.0 - moff;
            }
            wsync();
            SM_T(1);
            const double value = maximize(L, sm, budget, lane);
            SM_T(2);
            if (value <= tau + 1e-9) { st = SM_CONVERGED; break; }
            bool dup = false;
            for (int i = lane; i < m; i += 64) {
                bool eq = true;
                for (int w = 0; w < W; ++w) eq = eq && (cfg[i * W + w] == newc[w]);
                dup = dup || eq;
            }
            if (__any(dup)) { st = SM_DUPLICATE; break; }
            if (m >= cap) { st = SM_SINGULAR; break; }
            // z.push_back(0); distribution = z
            for (int i = lane; i <= m; i += 64) {
                const double zi = (i < m) ? z[i] : 0.0;
                if (i == m) z[i] = 0.0;
                dist[i] = zi;
            }
            // InvertAfterInsertion
            for (int i = lane; i <= m; i += 64) {
                double ri = 1.0;
                if (i > 0) {
                    int diff = 0;
                    for (int w = 0; w < W; ++w) diff += __popc(cfg[(i - 1) * W + w] ^ newc[w]);
                    ri = (double)(n - diff);
                }
                r[i] = ri;
            }
            wsync();
            SM_T(3);
            for (int j = lane; j <= m; j += 64) {
                double acc = 0.0;
#pragma unroll 8
                for (int i = 0; i <= m; ++i) {
                    const double ri = r[i];
                    const double t = acc + A[i * ld + j] * ri;
                    acc = (ri == 0.0) ? acc : t;  // terms with r_i == 0 are skipped (oracle C4)
                }
                d[j] = acc;
            }
            wsync();
            SM_T(4);
            double sden = (double)n;
#pragma unroll 8
            for (int j = 0; j <= m; ++j) sden -= r[j] * d[j];
            if (sden <= 1e-9 && sden >= -1e-9) { st = SM_SINGULAR; break; }
            const double invs = 1.0 / sden;
            SM_T(5);
            for (int j = lane; j <= m; j += 64) {  // lane owns column j; d[] is read-only here
                const double dj = d[j];
                double* __restrict__ col = A + j;          // A and d never overlap
                const double* __restrict__ dd = d;
#pragma unroll 8
                for (int i = 0; i <= m; ++i) col[i * ld] = col[i * ld] + invs * dd[i] * dj;
            }
            for (int i = lane; i <= m; i += 64) {
                const double v = -invs * d[i];
                A[i * ld + (m + 1)] = v;
                A[(m + 1) * ld + i] = v;
            }
            if (lane == 0) A[(m + 1) * ld + (m + 1)] = invs;
            for (int w = lane; w < W; w += 64) cfg[m * W + w] = newc[w];
            wsync();
            SM_T(6);
            {
                const double v = evaluate(L, sm, cfg + m * W);
                if (lane == 0) bsc[m + 1] = v;
            }
            ++m;
            changed = true;
            wsync();
            SM_T(7);
        } else {
            // blocking constraints: argmin of dist/(dist - z) over {z < dist}, first index on ties
            double best = INFINITY;
            int best_i = 0x7fffffff;
            bool exist = false;
            for (int i = lane; i < m; i += 64) {
                const double zi = z[i], pi = dist[i];
                if (zi >= pi) continue;
                if (zi < 0) exist = true;
                const double tmp = pi / (pi - zi);
                if (tmp < best) { best = tmp; best_i = i; }  // ascending i inside a lane
            }
            exist = __any(exist);
            if (!exist) {
                for (int i = lane; i < m; i += 64) dist[i] = z[i];
                changed = false;
                wsync();
                SM_T(8);
            } else {
                // wave argmin (value, then index)
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    const double ob = bcast_f64(best, (lane ^ o));
                    const int oi = __shfl(best_i, lane ^ o);
                    if (ob < best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
                }
                double alpha = best;
                const int blocking = best_i;
                if (alpha > 1.0) alpha = 1.0;
                if (alpha == 1.0) {
                    for (int i = lane; i < m; i += 64) dist[i] = z[i];
                } else {
                    for (int i = lane; i < m; i += 64) {
                        const double v = (1 - alpha) * dist[i] + alpha * z[i];
                        z[i] = v;
                        dist[i] = v;
                    }
                }
                // InvertAfterRemoval
                const int ri = blocking + 1;
                wsync();
                const double invs = A[ri * ld + ri];
                const double sdiv = 1.0 / invs;
                for (int i = lane; i <= m; i += 64)
                    if (i != ri) d[i - (i > ri ? 1 : 0)] = -sdiv * A[ri * ld + i];
                wsync();
                // in place with compaction: targets never pass the sources in row-major order
                for (int i = 0; i <= m; ++i) {
                    if (i == ri) continue;
                    const int kk = i - (i > ri ? 1 : 0);
                    const double dk = d[kk];
                    for (int j0 = 0; j0 <= m; j0 += 64) {
                        const int j = j0 + lane;
                        double v = 0.0;
                        const bool ok = (j <= m) && (j != ri);
                        // one wave: LDS executes its instructions in order, so every lane's read of
                        // this chunk precedes every lane's write below
                        if (ok) v = A[i * ld + j];
                        if (ok) {
                            const int l = j - (j > ri ? 1 : 0);
                            A[kk * ld + l] = v - invs * dk * d[l];
                        }
                    }
                }
                // erase entry `blocking` from cfg / z / dist / bsc (chunks ascending: read, then write)
                for (int i0 = blocking; i0 < m - 1; i0 += 64) {
                    const int i = i0 + lane;
                    const bool ok = i < m - 1;
                    double zv = 0.0, dv = 0.0, bv = 0.0;
                    uint32_t cw[4] = {0u, 0u, 0u, 0u};
                    if (ok) {
                        zv = z[i + 1]; dv = dist[i + 1]; bv = bsc[i + 2];
#pragma unroll
                        for (int w = 0; w < 4; ++w) if (w < W) cw[w] = cfg[(i + 1) * W + w];
                    }
                    wsync();
                    if (ok) {
                        z[i] = zv; dist[i] = dv; bsc[i + 1] = bv;
#pragma unroll
                        for (int w = 0; w < 4; ++w) if (w < W) cfg[i * W + w] = cw[w];
                    }
                    wsync();
                }
                --m;
                changed = true;
                wsync();
                SM_T(9);
            }
        }
    }
    wsync();

    // outputs
    for (int i = lane; i < cap * W; i += 64)
        aset_bits[s * cap * W + i] = (i < m * W) ? cfg[i] : 0u;
    for (int i = lane; i < cap; i += 64) p_out[s * cap + i] = (i < m) ? (float)dist[i] : 0.0f;
    if (inv_a) {
        double* o = inv_a + s * (int64_t)ld * ld;
        for (int i = 0; i <= m; ++i)
            for (int j = lane; j <= m; j += 64) o[i * ld + j] = A[i * ld + j];
    }
    if (lane == 0) {
        alen[s] = m;
        if (iters) iters[s] = it;
        if (status) status[s] = st;
    }
}

// dist_jacobian_vec (lvmhelpers/sparsemap.py:38-44): tmp = S dp with S = invA[1:,1:], out = M tmp
// restricted to the first n (on) coordinates.
__global__ __launch_bounds__(64) void sparsemap_bwd_kernel(
    const float* __restrict__ dp, const uint32_t* __restrict__ aset_bits,
    const int32_t* __restrict__ alen, const double* __restrict__ inv_a, int64_t B, int n,
    float* __restrict__ dscores)
{
    extern __shared__ __align__(16) char sm[];
    double* tmp = (double*)sm;                         // [cap]
    double* dpd = tmp + (n + 1);                       // [cap]
    uint32_t* cfg = (uint32_t*)(dpd + (n + 1));        // [cap*W]
    const int lane = threadIdx.x;
    const int64_t s = blockIdx.x;
    const int cap = n + 1, ld = n + 2, W = (n + 31) / 32;
    const int m = alen[s];
    const double* A = inv_a + s * (int64_t)ld * ld;
    for (int j = lane; j < m; j += 64) dpd[j] = (double)dp[s * cap + j];
    for (int i = lane; i < m * W; i += 64) cfg[i] = aset_bits[s * cap * W + i];
    __syncthreads();
    for (int i = lane; i < m; i += 64) {
        double acc = 0.0;
        for (int j = 0; j < m; ++j) acc += A[(i + 1) * ld + (j + 1)] * dpd[j];
        tmp[i] = acc;
    }
    __syncthreads();
    for (int k = lane; k < n; k += 64) {
        double acc = 0.0;
        for (int i = 0; i < m; ++i)
            if ((cfg[i * W + (k >> 5)] >> (k & 31)) & 1u) acc += tmp[i];
        dscores[s * n + k] = (float)acc;
    }
}

}  // namespace smlvm

using namespace smlvm;

extern "C" int smlvm_sparsemap_fwd(const float* scores, const double* init_scores, int32_t budget,
                                   int32_t max_iter, int64_t B, int32_t n, uint32_t* aset_bits,
                                   float* p, int32_t* alen, double* inv_a, int32_t* iters,
                                   int32_t* status, void* stream)
{
    if (B < 0 || n < 1 || max_iter < 0 || budget < 0) return SMLVM_ERR_INVALID;
    if (n > SMLVM_SPARSEMAP_MAX_BITS) return SMLVM_ERR_UNSUPPORTED;
    if (B == 0) return SMLVM_OK;
    if (!scores || !aset_bits || !p || !alen) return SMLVM_ERR_INVALID;
    const SmLayout L(n);
    static thread_local size_t configured = 0;
    if (L.total > 48 * 1024 && L.total > configured) {
        if (hipFuncSetAttribute((const void*)sparsemap_fwd_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)L.total) != hipSuccess) {
            g_last_hip_error = (int)hipGetLastError();
            return SMLVM_ERR_HIP;
        }
        configured = L.total;
    }
    hipLaunchKernelGGL(sparsemap_fwd_kernel, dim3((unsigned)B), dim3(64), L.total,
                       (hipStream_t)stream, scores, init_scores, budget, max_iter, B, n, aset_bits, p,
                       alen, inv_a, iters, status);
    return check_launch();
}

extern "C" int smlvm_sparsemap_bwd(const float* dp, const uint32_t* aset_bits, const int32_t* alen,
                                   const double* inv_a, int64_t B, int32_t n, float* dscores,
                                   void* stream)
{
    if (B < 0 || n < 1) return SMLVM_ERR_INVALID;
    if (n > SMLVM_SPARSEMAP_MAX_BITS) return SMLVM_ERR_UNSUPPORTED;
    if (B == 0) return SMLVM_OK;
    if (!dp || !aset_bits || !alen || !inv_a || !dscores) return SMLVM_ERR_INVALID;
    const int cap = n + 1, W = (n + 31) / 32;
    const size_t lds = sizeof(double) * 2 * cap + sizeof(uint32_t) * (size_t)cap * W;
    hipLaunchKernelGGL(sparsemap_bwd_kernel, dim3((unsigned)B), dim3(64), lds, (hipStream_t)stream,
                       dp, aset_bits, alen, inv_a, B, n, dscores);
    return check_launch();
}
