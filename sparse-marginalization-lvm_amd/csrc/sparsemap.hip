// Kernel 3: bit-vector SparseMAP -- active-set QP solver, one wavefront (= one workgroup of 64
// lanes) per sample, everything the solver touches resident in LDS.
//
// Replaces, for a whole batch, lvmhelpers/sparsemap.py:12-36 (PFactorGraph construction,
// init_active_set_from_scores, lpsmap GenericFactor::SolveQP, get_sparse_solution) with the linear
// oracles of lvmhelpers/bernoulli.h:14-35 (Bernoulli) and lvmhelpers/budget.h:24-85 (budget), and
// lvmhelpers/sparsemap.py:38-44 (dist_jacobian_vec) for the backward.
//
// The arithmetic follows oracle/sparsemap_oracle.c choice by choice (C1..C9 there) so that the
// active set -- which is path dependent -- comes out identical (C9, round 3: the drift monitor and
// polish mode, is in kkt_polish and at the solve of the main loop): every fp64 sum that feeds a
// decision adds the same numbers in the oracle's order (C1: the long sums as eight interleaved
// partial sums combined pairwise, the marginals and residual rows left to right), owned by ONE
// lane per output (lane c owns output c; sums that produce a single scalar are accumulated
// redundantly by all lanes from LDS broadcasts), there is no FMA contraction (this file is built
// with -ffp-contract=off; the two explicit fma() calls multiply by exactly 1.0 or 0.0, which makes
// them plain additions) and division is IEEE.  Parallelism comes from the 64 lanes owning
// different outputs, from the eight chains of a blocked sum, from the rank-one updates of the
// cached inverse KKT matrix, and from one workgroup per sample.
//
// The solver is latency bound (a lone wave per SIMD walks dependent fp64 additions), so the loops
// are written as 8-step batches whose LDS operands are fetched one batch ahead (ping-pong
// registers), and the blocked sums give every step of a batch its own chain.  Every vector that
// such a loop reads is kept zero-padded to a multiple of 8 (x + 0.0 == x bit for bit; the sums
// never hold -0.0), and rows of the inverse beyond the live block always hold finite numbers
// (zero-initialised, later only stale values), so padded steps are exact no-ops.
//
// LDS per sample (template LD = storage order of the inverse KKT matrix >= n+2): 8.LD^2 bytes
// dominate: 9.5 KB at n = 32, 141 KB at n = 128 (one sample per CU at a time).
#include "common.hpp"

#include <atomic>
#include <type_traits>

#include <math.h>

namespace smlvm {

enum { SM_CONVERGED = 0, SM_MAXITER = 1, SM_SINGULAR = 2, SM_DUPLICATE = 3, SM_STALLED = 4 };

typedef double dbl2 __attribute__((ext_vector_type(2)));

constexpr int round8(int v) { return (v + 7) & ~7; }

// LD: rows/columns of storage for the inverse KKT matrix (indices 0..m, m <= LD-1).
// WT: 32-bit words per stored configuration (>= ceil(n/32)).  NK: 64-lane chunks over variables.
template <int LD, int WT, int NK>
struct SmLayout {
    static constexpr int LDR = round8(LD);      // rows of A (padded so 8-row batches stay inside)
    static constexpr int NA = round8(LD);       // per-active-entry vectors
    static constexpr int NV = NK * 64;          // per-variable vectors
    static constexpr int CAPR = round8(LD - 1);  // stored configurations (rows of cfg)
    static constexpr size_t A = 0;
    static constexpr size_t z = A + sizeof(double) * (size_t)LDR * LD;
    static constexpr size_t dist = z + sizeof(double) * NA;
    static constexpr size_t bsc = dist + sizeof(double) * NA;
    static constexpr size_t d = bsc + sizeof(double) * NA;
    static constexpr size_t r = d + sizeof(double) * NA;
    // c0..c2, e: the linear oracle's arrays; t1 / prod (insertion only) share the first two
    static constexpr int NC = NV > NA ? NV : NA;
    static constexpr size_t c0 = r + sizeof(double) * NA;
    static constexpr size_t c1 = c0 + sizeof(double) * NC;
    static constexpr size_t c2 = c1 + sizeof(double) * NC;
    static constexpr size_t e = c2 + sizeof(double) * NV;
    static constexpr size_t t1 = c0;
    static constexpr size_t prod = c1;
    static constexpr size_t cfg = e + sizeof(double) * NV;
    static constexpr size_t newc = cfg + sizeof(uint32_t) * (size_t)CAPR * WT;
    static constexpr size_t total = (newc + sizeof(uint32_t) * 4 + 15) & ~(size_t)15;
};

__device__ __forceinline__ void wsync() { __syncthreads(); }  // single-wave workgroup

// Diagnostic build only (tools/sm_prof.hip defines SMLVM_SM_PROF): cycles per solver phase.
#ifdef SMLVM_SM_PROF
// per-sample phase times: accumulated in registers, written once at the end of the kernel
// (g_prof_out [B][16] device memory, accumulated over the tiers of a sample)
__device__ unsigned long long* g_prof_out;
#define SM_T(k)                                                                   \
    do {                                                                          \
        long long now_ = clock64();                                               \
        prof_acc_[k] += (unsigned long long)(now_ - t_last_);                     \
        t_last_ = now_;                                                           \
    } while (0)
#define SM_T0() unsigned long long prof_acc_[16] = {0}; long long t_last_ = clock64()
#define SM_TEND(s_)                                                               \
    do {                                                                          \
        if (threadIdx.x == 0 && g_prof_out)                                       \
            for (int k_ = 0; k_ < 16; ++k_) g_prof_out[(s_) * 16 + k_] += prof_acc_[k_]; \
    } while (0)
#else
#define SM_T(k) do { } while (0)
#define SM_T0() do { } while (0)
#define SM_TEND(s_) do { } while (0)
#endif

// Every lane holds the same bits; tell the compiler (scalar registers, scalar branches).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool uni(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
__device__ __forceinline__ double uni(double v) { return readlane_f64(v, 0); }

__device__ __forceinline__ double bcast_f64(double v, int lane)
{
    long long b = __builtin_bit_cast(long long, v);
    int lo = __shfl((int)(b & 0xffffffffll), lane);
    int hi = __shfl((int)(b >> 32), lane);
    b = ((long long)hi << 32) | (unsigned int)lo;
    return __builtin_bit_cast(double, b);
}

struct OpMinD { __device__ __forceinline__ double operator()(double a, double b) const { return fmin(a, b); } };
struct OpMinI { __device__ __forceinline__ int operator()(int a, int b) const { return a < b ? a : b; } };
// Wave-wide lexicographic minimum of (value, index) pairs, one pair per lane (index 0x7fffffff =
// the lane has no candidate; values are not NaN): two DPP reductions -- the smallest value, then
// the smallest index among the lanes that hold it -- instead of a six-step butterfly of LDS
// permutes.  Every lane gets the winner; index 0x7fffffff if no lane had a candidate.
__device__ __forceinline__ void wave_argmin(double& best, int& best_i)
{
    const double v = best_i == 0x7fffffff ? INFINITY : best;
    const double vmin = readlane_f64(group_reduce<64>(v, OpMinD{}), 0);
    const int cand = (best_i != 0x7fffffff && v == vmin) ? best_i : 0x7fffffff;
    best_i = __builtin_amdgcn_readfirstlane(group_reduce<64>(cand, OpMinI{}));
    best = vmin;
}

// 8 consecutive doubles from a 16-byte aligned LDS address (4 x ds_read_b128)
__device__ __forceinline__ void ld8(const double* p, double (&v)[8])
{
    const dbl2* q = reinterpret_cast<const dbl2*>(p);
    const dbl2 t0 = q[0], t1 = q[1], t2 = q[2], t3 = q[3];
    v[0] = t0.x; v[1] = t0.y; v[2] = t1.x; v[3] = t1.y;
    v[4] = t2.x; v[5] = t2.y; v[6] = t3.x; v[7] = t3.y;
}

// C1 of the oracle: a long sum is accumulated in eight interleaved partial sums (term j goes to
// partial j mod 8, each partial left to right) that are combined pairwise at the end -- eight
// independent chains per sum, so a step costs an issue slot, not the latency of a dependent
// v_add_f64 (a lone wave per SIMD has nothing else to hide it with).
__device__ __forceinline__ double sum8_get(const double (&p)[8])
{
    return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
}

// C1, sums that produce one scalar from a vector spread over the lanes: every lane brings the
// partial of the terms j = lane (mod 64); the 64 partials are combined by a balanced binary tree
// over adjacent pairs (quads, half rows, rows by DPP, then (r0 + r1) + (r2 + r3) across the rows --
// at mirrored lanes a + b and b + a are the same bits), the same bits in every lane.
__device__ __forceinline__ double wave_sum(double partial)
{
    return readlane_f64(group_reduce<64>(partial, OpSum{}), 0);
}

// sum_{j < cnt8} col[j*LD] * b[j] as a blocked sum (C1; product rounded, then added); b zero-padded.
template <int LD>
__device__ __forceinline__ double matvec_col(const double* col, const double* b, int cnt8)
{
    double part[8];
    double a0[8], b0[8], a1[8], b1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { part[u] = 0.0; a0[u] = col[u * LD]; }
    ld8(b, b0);
    int j0 = 0;
    for (;;) {
        int jn = j0 + 8 < cnt8 ? j0 + 8 : cnt8 - 8;
#pragma unroll
        for (int u = 0; u < 8; ++u) a1[u] = col[(jn + u) * LD];
        ld8(b + jn, b1);
#pragma unroll
        for (int u = 0; u < 8; ++u) { const double t = a0[u] * b0[u]; part[u] += t; }
        j0 += 8;
        if (j0 >= cnt8) break;
        jn = j0 + 8 < cnt8 ? j0 + 8 : cnt8 - 8;
#pragma unroll
        for (int u = 0; u < 8; ++u) a0[u] = col[(jn + u) * LD];
        ld8(b + jn, b0);
#pragma unroll
        for (int u = 0; u < 8; ++u) { const double t = a1[u] * b1[u]; part[u] += t; }
        j0 += 8;
        if (j0 >= cnt8) break;
    }
    return sum8_get(part);
}

// out[q] = sum_{j < cnt8} A[j][c_q] rhs[j] for this lane's columns c_q = lane + 64 q of the chunks
// that hold an output or padding to clear (q 64 <= cnt8, c_q < NA); columns beyond LD - 1 read the
// last one (their result is discarded).
template <int LD, int NA>
__device__ __forceinline__ void matvec_chunks(const double* A, const double* rhs, int cnt8, int lane,
                                              double (&out)[(NA + 63) / 64])
{
    constexpr int NQ = (NA + 63) / 64;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        out[q] = 0.0;
        if (q * 64 <= cnt8) {
            const int c = lane + 64 * q;
            out[q] = matvec_col<LD>(A + (c < LD ? c : LD - 1), rhs, cnt8);
        }
    }
}

__device__ __forceinline__ double f64_from_hi(int hi)
{
    return __builtin_bit_cast(double, (unsigned long long)(unsigned)hi << 32);
}

// mon[q] = sum_i [bit k of cfg_i] z_i, moff[q] = sum_i [bit k off] z_i for k = lane + 64 q, serial
// in i as UpdateMarginalsFromConfiguration is called (bernoulli.h:58-73); z zero-padded to m8.
template <int WT, int NK>
__device__ __forceinline__ void marginals(const uint32_t* cfg, const double* z, int m8, int lane,
                                          double (&mon)[NK], double (&moff)[NK])
{
    int wi[NK];
#pragma unroll
    for (int q = 0; q < NK; ++q) {
        int w = 2 * q + (lane >> 5);
        wi[q] = w < WT ? w : WT - 1;
        mon[q] = 0.0;
        moff[q] = 0.0;
    }
    const int sh = lane & 31;
    if (m8 <= 0) return;
    // operands of batch i0+8 are fetched while batch i0 is being added
    double zz[8], zn[8];
    uint32_t w[NK][8], wn[NK][8];
    ld8(z, zz);
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int u = 0; u < 8; ++u) w[q][u] = cfg[u * WT + wi[q]];
#pragma unroll 2
    for (int i0 = 0; i0 < m8; i0 += 8) {
        const int in = i0 + 8 < m8 ? i0 + 8 : m8 - 8;
        ld8(z + in, zn);
#pragma unroll
        for (int q = 0; q < NK; ++q)
#pragma unroll
            for (int u = 0; u < 8; ++u) wn[q][u] = cfg[(in + u) * WT + wi[q]];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int q = 0; q < NK; ++q) {
                const int msk = __builtin_amdgcn_sbfe(w[q][u], sh, 1);       // 0 or -1
                const int fh = msk & 0x3ff00000;                              // 1.0 or 0.0 (high word)
                const double f_on = f64_from_hi(fh), f_off = f64_from_hi(fh ^ 0x3ff00000);
                mon[q] = __builtin_fma(zz[u], f_on, mon[q]);                   // == mon + z or mon + 0
                moff[q] = __builtin_fma(zz[u], f_off, moff[q]);
            }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            zz[u] = zn[u];
#pragma unroll
            for (int q = 0; q < NK; ++q) w[q][u] = wn[q][u];
        }
    }
}

template <int LD, int WT, int NK>
struct Sm {
    using L = SmLayout<LD, WT, NK>;
    char* sm;
    int n, n8, W, lane;
    __device__ __forceinline__ double* A() const { return (double*)(sm + L::A); }
    __device__ __forceinline__ double* z() const { return (double*)(sm + L::z); }
    __device__ __forceinline__ double* dist() const { return (double*)(sm + L::dist); }
    __device__ __forceinline__ double* bsc() const { return (double*)(sm + L::bsc); }
    __device__ __forceinline__ double* d() const { return (double*)(sm + L::d); }
    __device__ __forceinline__ double* r() const { return (double*)(sm + L::r); }
    __device__ __forceinline__ double* t1() const { return (double*)(sm + L::t1); }
    __device__ __forceinline__ double* prod() const { return (double*)(sm + L::prod); }
    __device__ __forceinline__ double* c0() const { return (double*)(sm + L::c0); }
    __device__ __forceinline__ double* c1() const { return (double*)(sm + L::c1); }
    __device__ __forceinline__ double* c2() const { return (double*)(sm + L::c2); }
    __device__ __forceinline__ double* e() const { return (double*)(sm + L::e); }
    __device__ __forceinline__ uint32_t* cfg() const { return (uint32_t*)(sm + L::cfg); }
    __device__ __forceinline__ uint32_t* newc() const { return (uint32_t*)(sm + L::newc); }

    __device__ __forceinline__ void put_bits(int q, unsigned long long mask) const
    {
        if (lane == 0) {
            uint32_t* nc = newc();
            if (2 * q < WT) nc[2 * q] = (uint32_t)mask;
            if (2 * q + 1 < WT) nc[2 * q + 1] = (uint32_t)(mask >> 32);
        }
    }

    // Linear oracle on the per-lane potentials on[]/off[] (variable k = lane + 64 q) -> newc[]
    // (LDS bit words).  Returns the oracle's value and Evaluate(newc) under eta = [x; 0]
    // (bernoulli.h:38-54: sum of eta over the on-bits in index order).
    // budget <= 0: lvmhelpers/bernoulli.h:14-35; budget > 0: lvmhelpers/budget.h:24-85.
    __device__ __forceinline__ void maximize(const double (&on)[NK], const double (&off)[NK],
                                             const double (&eta)[NK], int budget, double& value,
                                             double& evalsum) const
    {
        double* a1 = c1();
        double* a2 = c2();
        if (budget <= 0) {
            // C1: term k goes to partial k mod 64 = this lane, in the order q = 0, 1, ...
            double pv = 0.0, pe = 0.0;
#pragma unroll
            for (int q = 0; q < NK; ++q) {
                const int k = lane + 64 * q;
                const bool valid = k < n;
                const bool bit = valid && (on[q] > off[q]);
                pv += valid ? (bit ? on[q] : off[q]) : 0.0;
                pe += bit ? eta[q] : 0.0;
                put_bits(q, __ballot(bit));
            }
            value = wave_sum(pv);
            evalsum = wave_sum(pe);
            wsync();   // newc
            return;
        }
        // budget factor
        double ev[NK];
        bool sel[NK];
        int num_active = 0;
        double p_off = 0.0, p_gain = 0.0;   // C1: partials of sum off_k and of the selected gains
#pragma unroll
        for (int q = 0; q < NK; ++q) {
            const int k = lane + 64 * q;
            const bool valid = k < n;
            ev[q] = on[q] - off[q];
            sel[q] = valid && (ev[q] > 0);
            p_off += valid ? off[q] : 0.0;
            p_gain += sel[q] ? ev[q] : 0.0;  // adding 0.0 for the inactive ones changes nothing
            num_active += __popcll(__ballot(sel[q]));
        }
        if (uni(num_active > budget)) {
            // More positive gains than the budget admits: the `budget` largest are kept (sort of
            // (-e_i, i) ascending, budget.h:49-76).  Only the positives compete -- every one of
            // them sorts before every e_i <= 0 -- so they are compacted first (in index order:
            // comparing slots is comparing indices) and ranked among themselves: the pair loop
            // runs over num_active entries instead of n.
            const int P = uni(num_active), P8 = round8(P);
            int slot[NK];
            {
                int base = 0;
#pragma unroll
                for (int q = 0; q < NK; ++q) {
                    const unsigned long long mk = __ballot(sel[q]);
                    slot[q] = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0));
                    base += __popcll(mk);
                }
            }
            double* ee = e();
#pragma unroll
            for (int q = 0; q < NK; ++q) {
                const int k = lane + 64 * q;
                a1[k] = 0.0;                                  // (ranks cover 0..P-1, the rest stays zero)
                if (sel[q]) ee[slot[q]] = ev[q];
                if (k >= P && k < P8) ee[k] = -INFINITY;      // padding never outranks anything
            }
            wsync();
            int rank[NK];
#pragma unroll
            for (int q = 0; q < NK; ++q) rank[q] = 0;
            // first with "greater" alone (one compare + one add-with-carry per pair): exact unless
            // two gains are equal, in which case two variables get the same rank
            for (int j0 = 0; j0 < P8; j0 += 8) {
                double ej[8];
                ld8(ee + j0, ej);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
#pragma unroll
                    for (int q = 0; q < NK; ++q) rank[q] += (ej[u] > ev[q]) ? 1 : 0;
                }
            }
            // ranks of the positives are distinct iff there was no tie: every one claims its slot
            // and looks who holds it afterwards
            int* claim = reinterpret_cast<int*>(a2);
            bool clash = false;
#pragma unroll
            for (int q = 0; q < NK; ++q)
                if (sel[q]) claim[rank[q]] = slot[q];
            wsync();
#pragma unroll
            for (int q = 0; q < NK; ++q)
                if (sel[q]) clash = clash || (claim[rank[q]] != slot[q]);
            wsync();
            if (__any(clash)) {   // rare: the full rule with the index tie-break
#pragma unroll
                for (int q = 0; q < NK; ++q) rank[q] = 0;
                for (int j0 = 0; j0 < P8; j0 += 8) {
                    double ej[8];
                    ld8(ee + j0, ej);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int j = j0 + u;
#pragma unroll
                        for (int q = 0; q < NK; ++q)
                            rank[q] += ((ej[u] > ev[q]) || (ej[u] == ev[q] && j < slot[q])) ? 1 : 0;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NK; ++q) {
                const bool pos = sel[q];
                sel[q] = pos && (rank[q] < budget);
                if (pos) a1[rank[q]] = sel[q] ? ev[q] : 0.0;  // sorted, zero beyond the selection
            }
            wsync();
            // the chosen gains are summed in RANK order: rank r goes to partial r mod 64
            p_gain = 0.0;
#pragma unroll
            for (int q = 0; q < NK; ++q) p_gain += a1[lane + 64 * q];
        }
        double pe = 0.0;
#pragma unroll
        for (int q = 0; q < NK; ++q) {
            pe += sel[q] ? eta[q] : 0.0;
            put_bits(q, __ballot(sel[q]));
        }
        const double v_off = wave_sum(p_off), v_gain = wave_sum(p_gain);
        value = v_off + v_gain;
        evalsum = wave_sum(pe);
        wsync();   // newc; a1 / e are reused by the next call
    }

    // Linear oracle of the binary sequence factor (lvmhelpers/sequence_binary.h:11-38 on
    // sequence.h:66-139; oracle/sparsemap_oracle.c maximize_seq): Viterbi over n positions with two
    // states, node scores on[] / off[], edge score trs[k] = t[k-1] for the (1 -> 1) transition into
    // position k (use_trans = false: zero edge potentials, InitActiveSetFromScores).  The recursion
    // is serial by nature; every lane walks it on LDS broadcasts and gets the same bits.  Returns
    // the value and Evaluate(newc) (sequence.h:142-166): node and edge scores interleaved in
    // position order, under eta and the TRUE edge potentials.
    __device__ __forceinline__ void maximize_seq(const double (&on)[NK], const double (&off)[NK],
                                                 const double (&eta)[NK], const double (&trs)[NK],
                                                 bool use_trans, double& value, double& evalsum) const
    {
        double* a0 = c0();
        double* a1 = c1();
        double* a2 = c2();
#pragma unroll
        for (int q = 0; q < NK; ++q) {
            const int k = lane + 64 * q;
            const bool valid = k < n;
            a0[k] = valid ? on[q] : 0.0;
            a1[k] = valid ? off[q] : 0.0;
            a2[k] = (valid && use_trans) ? trs[q] : 0.0;
        }
        wsync();
        // position 0 as a step from v0 = v1 = 0 with trs[0] = 0: 0.0 + x == x + 0.0 bit for bit
        double v0 = 0.0, v1 = 0.0;
        unsigned long long f0[NK], f1[NK];
#pragma unroll
        for (int q = 0; q < NK; ++q) { f0[q] = 0ull; f1[q] = 0ull; }
        // eight positions; TAIL: the batch that holds position n - 1 and padding (selects keep the
        // padded steps from moving the values; full batches do without them)
        auto batch = [&](int i0, auto tail) {
            constexpr bool TAIL = decltype(tail)::value;
            double o[8], f[8], t[8];
            ld8(a0 + i0, o);
            ld8(a1 + i0, f);
            ld8(a2 + i0, t);
            unsigned long long m0 = 0ull, m1 = 0ull;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double c00 = v0, c10 = v1;                   // into state 0 (edge score 0)
                const bool b0 = (__ballot(c10 > c00) & 1ull) != 0ull;   // lane 0 is always active
                const double best0 = b0 ? c10 : c00;
                const double c01 = v0, c11 = v1 + t[u];            // into state 1
                const bool b1 = (__ballot(c11 > c01) & 1ull) != 0ull;
                const double best1 = b1 ? c11 : c01;
                const double nv0 = best0 + f[u], nv1 = best1 + o[u];
                if constexpr (TAIL) {
                    const bool live = i0 + u < n;
                    v0 = live ? nv0 : v0;
                    v1 = live ? nv1 : v1;
                } else {
                    v0 = nv0;
                    v1 = nv1;
                }
                m0 |= (unsigned long long)b0 << u;
                m1 |= (unsigned long long)b1 << u;
            }
#pragma unroll
            for (int q = 0; q < NK; ++q)
                if ((i0 >> 6) == q) { f0[q] |= m0 << (i0 & 63); f1[q] |= m1 << (i0 & 63); }
        };
        {
            int i0 = 0;
            for (; i0 + 8 <= n; i0 += 8) batch(i0, std::false_type{});
            if (i0 < n) batch(i0, std::true_type{});
        }
        const double e0 = v0, e1 = v1;
        bool state = (__ballot(e1 > e0) & 1ull) != 0ull;
        value = state ? e1 : e0;
        unsigned long long bits[NK];
#pragma unroll
        for (int q = 0; q < NK; ++q) bits[q] = 0ull;
#pragma unroll
        for (int q = NK - 1; q >= 0; --q) {
            const int top = (n - 64 * q) < 64 ? (n - 64 * q) : 64;   // positions of this word
            for (int b = top - 1; b >= 0; --b) {
                if (state) bits[q] |= 1ull << b;
                if (b + 64 * q > 0) state = ((state ? f1[q] : f0[q]) >> b) & 1ull;
            }
        }
        wsync();   // the recursion's reads of a2 are done before it is reused below
        // Evaluate: ev[2k] = node score of position k, ev[2k+1] = edge score into position k
        double* ev = a2;   // c2 and e are contiguous: 2 NV doubles
#pragma unroll
        for (int q = 0; q < NK; ++q) {
            const int k = lane + 64 * q;
            const bool valid = k < n;
            const bool bit = valid && ((bits[q] >> lane) & 1ull);
            bool prev;
            if (lane > 0) prev = (bits[q] >> (lane - 1)) & 1ull;
            else prev = q > 0 ? ((bits[q > 0 ? q - 1 : 0] >> 63) & 1ull) : false;
            ev[2 * k] = bit ? eta[q] : 0.0;
            ev[2 * k + 1] = (bit && prev) ? trs[q] : 0.0;
            put_bits(q, bits[q]);
        }
        wsync();
        // C1: term j of the interleaved sequence goes to partial j mod 64
        double pe = 0.0;
#pragma unroll
        for (int t = 0; t < 2 * NK; ++t) pe += ev[lane + 64 * t];
        evalsum = wave_sum(pe);
        wsync();
    }
};

// ---- C7 of oracle/sparsemap_oracle.c: a convergence is only accepted on an accurate KKT solution
constexpr double kKktTol = 1e-10;      // accepted residual (any row)
constexpr double kKktRejudge = 1e-8;   // a starting residual above this voids the convergence decision
constexpr int kKktPolish = 2;          // refinement steps per attempt
constexpr int kKktMaxRebuilds = 3;     // recomputations of the inverse per sample (C7 and C9 together)
constexpr double kKktGiveUp = 1e-8;    // a polish that ends above this cannot be judged: STALLED
constexpr double kSingularTol = 1e-6;  // C11: |s| below this is a singular insertion
constexpr double kDriftTol = 1e-11;    // C9: |1 - sum z| above this turns polish mode on
constexpr int kKktMaxRejudges = 2;     // re-judgements per sample
constexpr int kKktPatience = 64;       // passes a re-judged sample gets to converge

// res = [1; b] - K [tau; z], K = [[0,1^T],[1,G]], G_ij = common values of configurations i, j.
// Lane i + 64 q owns row i + 64 q: a serial left-to-right sum over j (product rounded, then
// added), exactly the oracle's kkt_residual.  res goes to r[] (zero beyond m: it is the right-hand
// side of the refinement's matvec).  Returns max_i |res_i|, the same bits in every lane.
template <int WT, int NA>
__device__ __forceinline__ double kkt_residual(const double* z, double tau, const double* bsc,
                                               const uint32_t* cfg, int m, int n, double* r, int lane)
{
    // z is zero beyond m and the configuration rows beyond m hold finite words, so the serial sums
    // run over round8(m) terms in 8-step batches (a + 0.0 == a) with the operands of a batch
    // fetched together; the additions themselves stay strictly left to right.
    const int m8 = round8(m);
    double worst = 0.0;
#pragma unroll
    for (int q = 0; q * 64 < NA; ++q) {
        const int i = lane + 64 * q;
        if (q * 64 <= m && i < NA) {
            double res = 0.0;
            if (i == 0) {
                double acc = 0.0;
                for (int j0 = 0; j0 < m8; j0 += 8) {
                    double zz[8];
                    ld8(z + j0, zz);
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += zz[u];
                }
                res = 1.0 - acc;
            } else if (i <= m) {
                uint32_t ci[WT];
#pragma unroll
                for (int w = 0; w < WT; ++w) ci[w] = cfg[(i - 1) * WT + w];
                double a = tau;
                for (int j0 = 0; j0 < m8; j0 += 8) {
                    double zz[8], t[8];
                    ld8(z + j0, zz);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        int diff = 0;
#pragma unroll
                        for (int w = 0; w < WT; ++w) diff += __popc(ci[w] ^ cfg[(j0 + u) * WT + w]);
                        t[u] = (double)(n - diff) * zz[u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) a += t[u];
                }
                res = bsc[i] - a;
            }
            r[i] = res;
            worst = fmax(worst, fabs(res));
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) worst = fmax(worst, bcast_f64(worst, lane ^ o));
    wsync();
    return uni(worst);
}

// A (indices 0..m) := K^-1 by in-place Gauss-Jordan elimination without pivoting, pivots in the
// order 1..m, 0 (the oracle's rebuild_inverse).  Every element update is v - c_i * (r_j * piv):
// element-wise, no sums, so the lanes (one column each) produce the oracle's bits.  colv / rowv:
// LDS scratch of round8(m + 1) doubles each.
template <int LD, int WT>
__device__ __forceinline__ void rebuild_inverse(double* A, const uint32_t* cfg, int m, int n,
                                                double* colv, double* rowv, int lane)
{
    for (int i = 0; i <= m; ++i) {
        for (int j = lane; j <= m; j += 64) {
            double v;
            if (i == 0) v = (j == 0) ? 0.0 : 1.0;
            else if (j == 0) v = 1.0;
            else {
                int diff = 0;
#pragma unroll
                for (int w = 0; w < WT; ++w) diff += __popc(cfg[(i - 1) * WT + w] ^ cfg[(j - 1) * WT + w]);
                v = (double)(n - diff);
            }
            A[i * LD + j] = v;
        }
    }
    wsync();
    const int m8 = round8(m + 1);
    for (int step = 0; step <= m; ++step) {
        const int k = step < m ? step + 1 : 0;
        const double piv = uni(1.0 / A[k * LD + k]);
        // column k and the scaled row k; colv gets zeros at k and in the padding, so that the
        // update below needs no conditions (v - 0 * r_j == v) and runs in 8-row batches
        for (int i = lane; i < m8; i += 64) {
            const bool in = i <= m;
            colv[i] = (in && i != k) ? A[i * LD + k] : 0.0;
            rowv[i] = in ? A[k * LD + i] * piv : 0.0;
        }
        wsync();
        for (int j = lane; j <= m; j += 64) {
            if (j == k) continue;
            const double rj = rowv[j];
            double* col = A + j;
            for (int i0 = 0; i0 < m8; i0 += 8) {   // rows up to m8 - 1 < LDR exist (stale, finite)
                double a[8], c[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = col[(i0 + u) * LD];
                ld8(colv + i0, c);
#pragma unroll
                for (int u = 0; u < 8; ++u) { const double t = c[u] * rj; a[u] = a[u] - t; }
#pragma unroll
                for (int u = 0; u < 8; ++u) col[(i0 + u) * LD] = a[u];
            }
        }
        wsync();
        for (int i = lane; i <= m; i += 64) {
            if (i == k) continue;
            A[k * LD + i] = rowv[i];
            A[i * LD + k] = -colv[i] * piv;
        }
        if (lane == 0) A[k * LD + k] = piv;
        wsync();
    }
}

// SEQ: the binary sequence factor (sequence_smap, lvmhelpers/sparsemap.py:47-87,130-148): trans
// [B, n-1] fp32 edge potentials of the (1 -> 1) transitions; Viterbi oracle, edge scores in
// Evaluate, and choice C8 of oracle/sparsemap_oracle.c (vertex swap on a singular insertion).
template <int LD, int WT, int NK, bool SEQ>
__global__ __launch_bounds__(64) void sparsemap_fwd_kernel(
    const float* __restrict__ scores, const float* __restrict__ trans,
    const double* __restrict__ init_scores, int budget,
    int max_iter, int64_t B, int n, uint32_t* __restrict__ aset_bits, float* __restrict__ p_out,
    int32_t* __restrict__ alen, double* __restrict__ a_out, int32_t* __restrict__ iters,
    int32_t* __restrict__ status, int resume, int32_t* __restrict__ ws_state,
    int32_t* __restrict__ ws_it, double* __restrict__ ws_tau, double* __restrict__ ws_vec)
{
    // Tiers (n > 64): the first launch runs every sample with a small LD (several samples per CU);
    // a sample whose active set outgrows it parks its state (a_out, aset_bits, alen, ws_*) and a
    // later launch with a larger LD (resume = 1) picks it up at the top of the same iteration.
    using L = SmLayout<LD, WT, NK>;
    extern __shared__ __align__(16) char smem[];
    Sm<LD, WT, NK> S;
    S.sm = smem;
    S.n = n;
    S.n8 = round8(n);
    S.W = (n + 31) / 32;
    S.lane = threadIdx.x;
    const int lane = threadIdx.x;
    const int64_t s = blockIdx.x;
    const int cap = n + 1, W = S.W, ldo = n + 2;
    if (resume && (ws_state[s] & 1) == 0) return;   // whole wave: finished in an earlier tier
    double* A = S.A();
    double* z = S.z();
    double* dist = S.dist();
    double* bsc = S.bsc();
    double* d = S.d();
    double* r = S.r();
    double* t1 = S.t1();
    uint32_t* cfg = S.cfg();
    uint32_t* newc = S.newc();

    // zero everything: padded steps of the serial loops rely on zeros / finite values
    {
        dbl2* all = reinterpret_cast<dbl2*>(smem);
        const dbl2 zero2 = {0.0, 0.0};
        for (int i = lane; i < (int)(L::total / 16); i += 64) all[i] = zero2;
    }
    double eta[NK], on[NK], off[NK];
    double trs[NK];   // SEQ: trs[q] = t[k-1], the edge potential into position k = lane + 64 q
#pragma unroll
    for (int q = 0; q < NK; ++q) {
        const int k = lane + 64 * q;
        eta[q] = 0.0; on[q] = 0.0; off[q] = 0.0;
        trs[q] = 0.0;
        if constexpr (SEQ) {
            if (k >= 1 && k < n) trs[q] = (double)trans[s * (n - 1) + k - 1];
        }
        if (k < n) {
            const double x = (double)scores[s * n + k];
            eta[q] = x;
            if (init_scores) {  // InitActiveSetFromScores: Maximize(init)
                on[q] = init_scores[s * 2 * n + k];
                off[q] = init_scores[s * 2 * n + n + k];
            } else {            // SolveQP on an empty active set: Maximize(eta)
                on[q] = x;
                off[q] = 0.0;
            }
        }
    }
    wsync();
    int m = 1, it = 0, rebuilds = 0, rejudges = 0, last_rejudge = 0;
    bool changed = true, polish = false;   // polish: C9, sticky
    double tau = 0.0;
    if (!resume) {
        double value, evalsum;
        if constexpr (SEQ) S.maximize_seq(on, off, eta, trs, init_scores == nullptr, value, evalsum);
        else S.maximize(on, off, eta, budget, value, evalsum);
        if (lane < WT) cfg[lane] = newc[lane];
        if (lane == 0) {
            dist[0] = 1.0;
            A[0 * LD + 0] = -(double)n; A[0 * LD + 1] = 1.0; A[1 * LD + 0] = 1.0; A[1 * LD + 1] = 0.0;
            bsc[0] = 1.0;
            bsc[1] = evalsum;
        }
    } else {
        m = uni(alen[s]);
        it = uni(ws_it[s]);
        polish = uni((ws_state[s] >> 1) & 1) != 0;
        rebuilds = uni((ws_state[s] >> 4) & 15);
        rejudges = uni((ws_state[s] >> 8) & 15);
        last_rejudge = uni(ws_state[s] >> 12);
        tau = uni(ws_tau[s]);
        changed = false;   // parked at the top of an insertion pass
        const double* ai = a_out + s * (int64_t)ldo * ldo;
        for (int i = 0; i <= m; ++i)
            for (int j = lane; j <= m; j += 64) A[i * LD + j] = ai[i * ldo + j];
        const double* vi = ws_vec + s * (int64_t)3 * ldo;
        for (int i = lane; i <= m; i += 64) {
            if (i < m) { z[i] = vi[i]; dist[i] = vi[ldo + i]; }
            bsc[i] = vi[2 * ldo + i];
        }
        for (int i = lane; i < m * W; i += 64) {
            const int row = i / W, w = i - row * W;
            cfg[row * WT + w] = aset_bits[s * cap * W + i];
        }
    }
    wsync();

    SM_T0();
    int st = SM_MAXITER;
    bool overflow = false;
    // refined / pending are never set when a sample is parked (that happens in an insertion pass,
    // after they were consumed); judged, polish and the rebuild / re-judgement counts travel in
    // ws_state
    bool judged = false, refined = false, pending = false;
    if (resume) judged = uni((ws_state[s] >> 2) & 1) != 0;
    // [tau; z] (+)= invA^T rhs: lane c owns output c, blocked over j like the oracle (C1)
    auto kkt_apply = [&](const double* rhs, bool accumulate) __attribute__((always_inline)) {
        const int cnt8 = round8(m + 1);
        double t0 = 0.0;
        double mv[(L::NA + 63) / 64];
        matvec_chunks<LD, L::NA>(A, rhs, cnt8, lane, mv);
#pragma unroll
        for (int q = 0; q * 64 < L::NA; ++q) {
            const int c = lane + 64 * q;
            if (q * 64 <= cnt8) {   // uniform: chunks that hold an output or padding to clear
                const double acc = mv[q];
                if (c == 0) t0 = acc;
                else if (c - 1 < L::NA) {
                    if (accumulate) { if (c <= m) z[c - 1] = z[c - 1] + acc; }
                    else z[c - 1] = (c <= m) ? acc : 0.0;
                }
            }
        }
        tau = accumulate ? uni(tau + t0) : uni(t0);
        wsync();
    };
    // C7 / C9: the polish sequence on the current solution (the oracle's kkt_polish): residual, up
    // to kKktPolish refinement steps, a rebuild if that does not get there; returns the residual it
    // ends with (rho0: the one it started from).  Uses r (residual), d and t1 (rebuild scratch):
    // free outside an insertion.
    auto kkt_polish = [&](double& rho0, int min_rounds) __attribute__((always_inline)) -> double {
        rho0 = kkt_residual<WT, L::NA>(z, tau, bsc, cfg, m, n, r, lane);
        double rho = rho0;
        // min_rounds = 1 (C10): one refinement step whatever the residual
        for (int round = 0; round < kKktPolish && (rho > kKktTol || round < min_rounds); ++round) {
            kkt_apply(r, true);   // iterative refinement: [dtau; dz] = invA^T res
            rho = kkt_residual<WT, L::NA>(z, tau, bsc, cfg, m, n, r, lane);
        }
        if (rho > kKktTol && rebuilds < kKktMaxRebuilds) {
            // the cached inverse is too far off for the refinement to converge
            rebuild_inverse<LD, WT>(A, cfg, m, n, d, t1, lane);
            ++rebuilds;
            kkt_apply(bsc, false);
            rho = kkt_residual<WT, L::NA>(z, tau, bsc, cfg, m, n, r, lane);
            for (int round = 0; round < kKktPolish && rho > kKktTol; ++round) {
                kkt_apply(r, true);
                rho = kkt_residual<WT, L::NA>(z, tau, bsc, cfg, m, n, r, lane);
            }
        }
        return rho;
    };
    // InvertAfterRemoval(idx) and erase entry idx of the active set (m decreases by one)
    auto remove_entry = [&](int idx) __attribute__((always_inline)) {
        // InvertAfterRemoval
        const int ri = idx + 1;
        wsync();
        const double invs = uni(A[ri * LD + ri]);
        const double sdiv = uni(1.0 / invs);
        for (int i = lane; i <= m; i += 64)
            if (i != ri) d[i - (i > ri ? 1 : 0)] = -sdiv * A[ri * LD + i];
        wsync();
        // in place with compaction: targets never pass the sources in row-major order.
        // Rows go in batches of 8, all column chunks together: all reads of a batch are issued
        // before its writes (one wave: LDS executes in order), and a batch only writes rows it has
        // already read.  (Batches of 4 rows, chunk by chunk, exposed the LDS round trip 3 m / 4
        // times per removal: 30 % of the run time of the samples that zigzag to the iteration limit.)
        constexpr int RB = 8, NQA = (LD + 63) / 64;
        bool okj[NQA];
        int lq[NQA];
        double dl[NQA];
#pragma unroll
        for (int q = 0; q < NQA; ++q) {
            const int j = lane + 64 * q;
            okj[q] = (j <= m) && (j != ri);
            lq[q] = j - (j > ri ? 1 : 0);
            dl[q] = okj[q] ? d[lq[q]] : 0.0;
        }
        for (int i0 = 0; i0 <= m; i0 += RB) {
            double v[NQA][RB], dk[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int i = i0 + u;
                const int ic = i <= m ? i : m;
                const int kk = ic - (ic > ri ? 1 : 0);
                dk[u] = d[kk];
#pragma unroll
                for (int q = 0; q < NQA; ++q)
                    if (q * 64 <= m) v[q][u] = okj[q] ? A[ic * LD + lane + 64 * q] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int i = i0 + u;
                if (i <= m && i != ri) {
                    const int kk = i - (i > ri ? 1 : 0);
#pragma unroll
                    for (int q = 0; q < NQA; ++q)
                        if (q * 64 <= m && okj[q]) A[kk * LD + lq[q]] = v[q][u] - invs * dk[u] * dl[q];
                }
            }
        }
        // erase entry `idx` from cfg / z / dist / bsc (chunks ascending: read, then write)
        for (int i0 = idx; i0 < m; i0 += 64) {
            const int i = i0 + lane;
            const bool ok = i < m - 1;
            const bool last = i == m - 1;   // vacated slot: back to zero (padding invariant)
            double zv = 0.0, dv = 0.0, bv = 0.0;
            uint32_t cw[WT];
#pragma unroll
            for (int w = 0; w < WT; ++w) cw[w] = 0u;
            if (ok) {
                zv = z[i + 1]; dv = dist[i + 1]; bv = bsc[i + 2];
#pragma unroll
                for (int w = 0; w < WT; ++w) cw[w] = cfg[(i + 1) * WT + w];
            }
            wsync();
            if (ok || last) {
                z[i] = zv; dist[i] = dv; bsc[i + 1] = bv;
#pragma unroll
                for (int w = 0; w < WT; ++w) cfg[i * WT + w] = cw[w];
            }
            wsync();
        }
        --m;
    };
    // r = [1; CountCommonValues(cfg_i, new)]; true if the new configuration is already active
    auto common_with_new = [&]() __attribute__((always_inline)) -> bool {
        bool dup = false;
#pragma unroll
        for (int q = 0; q * 64 < L::NA; ++q) {
            const int i = lane + 64 * q;   // KKT index: 0 -> 1.0, i >= 1 -> configuration i-1
            if (i < L::NA) {
                double ri = 0.0;
                if (i == 0) ri = 1.0;
                else if (i <= m) {
                    int diff = 0;
#pragma unroll
                    for (int w = 0; w < WT; ++w) diff += __popc(cfg[(i - 1) * WT + w] ^ newc[w]);
                    dup = dup || (diff == 0);
                    ri = (double)(n - diff);
                }
                r[i] = ri;
            }
        }
        return __any(dup) != 0;
    };
    // d = invA r (zero beyond m) -- in polish mode refined by one step on the exact residual
    // r - K d (C11) --; returns s = n - sum_j r_j d_j
    auto insertion_direction = [&]() __attribute__((always_inline)) -> double {
        wsync();
        const int cnt8 = round8(m + 1);
        double mv[(L::NA + 63) / 64];
        double dl[(L::NA + 63) / 64];
        matvec_chunks<LD, L::NA>(A, r, cnt8, lane, mv);
#pragma unroll
        for (int q = 0; q * 64 < L::NA; ++q) {
            const int j = lane + 64 * q;
            dl[q] = 0.0;
            if (q * 64 <= cnt8 && j < L::NA) {
                dl[q] = (j <= m) ? mv[q] : 0.0;
                d[j] = dl[q];
            }
        }
        if (polish) {
            // the residual of kkt_residual with [tau; z] = d and right-hand side r; z = d[1..] is
            // copied to an aligned, zero-padded vector (prod), the residual goes to t1
            double* zz = S.prod();
            wsync();
#pragma unroll
            for (int q = 0; q * 64 < L::NA; ++q) {
                const int j = lane + 64 * q;
                if (q * 64 <= cnt8 && j < L::NA) zz[j] = (j < m) ? d[j + 1] : 0.0;
            }
            const double d0 = uni(d[0]);
            wsync();
            kkt_residual<WT, L::NA>(zz, d0, r, cfg, m, n, t1, lane);
            matvec_chunks<LD, L::NA>(A, t1, cnt8, lane, mv);
#pragma unroll
            for (int q = 0; q * 64 < L::NA; ++q) {
                const int j = lane + 64 * q;
                if (q * 64 <= cnt8 && j < L::NA && j <= m) {
                    dl[q] = dl[q] + mv[q];
                    d[j] = dl[q];
                }
            }
        }
        double prd = 0.0;
#pragma unroll
        for (int q = 0; q * 64 < L::NA; ++q) {
            const int j = lane + 64 * q;
            if (q * 64 <= cnt8 && j < L::NA) prd += r[j] * dl[q];   // r is 0 beyond m
        }
        wsync();
        SM_T(4);
        const double acc = (double)n - wave_sum(prd);   // C1: s = n - sum_j r_j d_j, partial j mod 64
        return acc;
    };
    for (; it < max_iter; ++it) {
        bool same = true;
        if (rejudges > 0 && it - last_rejudge > kKktPatience) { st = SM_STALLED; break; }   // C7
        if (changed) {
            // [tau; z] = invA^T b, b = [1; Evaluate(cfg_i)]
            kkt_apply(bsc, false);
            if (!polish) {   // C9: drift monitor, row 0 of the KKT system; term j -> partial j mod 64
                double pz = 0.0;
#pragma unroll
                for (int q = 0; q * 64 < L::NA; ++q)
                    if (q * 64 < m && lane + 64 * q < L::NA) pz += z[lane + 64 * q];   // zero beyond m
                polish = uni(fabs(1.0 - wave_sum(pz)) > kDriftTol);
            }
            SM_T(0);
            same = false;
        }
        if ((changed && polish) || pending) {
            // C9: every solve in polish mode; C10: the re-judgement asked for by the last pass (one
            // refinement step at least).  A polish that ends above kKktGiveUp (or NaN) cannot be
            // judged: STALLED, last feasible p.
            double rho0;
            const bool failed = !uni(kkt_polish(rho0, pending ? 1 : 0) <= kKktGiveUp);
            SM_T(7);
            if (failed) { st = SM_STALLED; break; }
            if (pending) { refined = true; pending = false; }
        }
        if (refined) { refined = false; same = false; }   // C7: z was re-solved in the last pass
        if (same) {
            double mon[NK], moff[NK];
            marginals<WT, NK>(cfg, z, round8(m), lane, mon, moff);
#pragma unroll
            for (int q = 0; q < NK; ++q) {
                on[q] = eta[q] - mon[q];
                off[q] = 0.0 - moff[q];
            }
            SM_T(1);
            double value, evalsum;
            if constexpr (SEQ) S.maximize_seq(on, off, eta, trs, true, value, evalsum);
            else S.maximize(on, off, eta, budget, value, evalsum);
            SM_T(2);
            if (uni(value <= tau + 1e-9)) {
                // C7: accept only an accurate solution of the KKT system
                double rho0;
                if (!uni(kkt_polish(rho0, 0) <= kKktGiveUp)) { st = SM_STALLED; break; }
                if (rho0 > kKktRejudge && !judged) {
                    // the solution moved by more than the convergence tolerance: the next passes
                    // judge the polished one (blocking test, then optimality)
                    if (rejudges >= kKktMaxRejudges) { st = SM_STALLED; break; }
                    ++rejudges;
                    last_rejudge = it;
                    judged = true;
                    refined = true;
                    polish = true;   // C9
                    SM_T(7);
                    continue;
                }
                for (int i = lane; i < m; i += 64) dist[i] = z[i];
                wsync();
                SM_T(7);
                st = SM_CONVERGED;
                break;
            }
            // SameConfiguration against every active entry + r = [1; CountCommonValues(cfg_i, new)]
            if (common_with_new()) {
                if (judged) { st = SM_DUPLICATE; break; }
                // C10: polish the solution (one refinement step at least) and judge again
                polish = true;
                judged = true;
                pending = true;   // polished at the top of the next pass
                continue;
            }
            if constexpr (!SEQ) {
                if (m >= cap) { st = SM_SINGULAR; break; }
                if (m + 2 > LD) { overflow = true; break; }   // next tier redoes this pass
            }
            // z.push_back(0); distribution = z
            for (int i = lane; i <= m; i += 64) {
                const double zi = (i < m) ? z[i] : 0.0;
                if (i == m) z[i] = 0.0;
                dist[i] = zi;
            }
            wsync();
            SM_T(3);
            // InvertAfterInsertion: d = invA r, s = n - r.d
            double sden = insertion_direction();
            double w_new = 0.0;   // weight the new vertex starts with
            if (uni(sden <= kSingularTol && sden >= -kSingularTol)) {
                if constexpr (!SEQ) {
                    if (judged) { st = SM_SINGULAR; break; }
                    // C10 (see the oracle's header)
                    polish = true;
                    judged = true;
                    pending = true;
                    continue;
                }
                // C8 (sequence factor only): the new vertex is an affine combination of the active
                // ones in variable space, y_new = sum_i c_i y_i with c = d[1..m].  Its score is
                // higher than the same combination of theirs, so weight moves onto it along
                // p_i -= theta c_i (marginals and sum p unchanged) until the first active vertex
                // reaches zero (first index on ties); that vertex leaves, the new one enters with
                // weight theta.
                double best = INFINITY;
                int best_i = 0x7fffffff;
                for (int i = lane; i < m; i += 64) {
                    const double di = d[i + 1];
                    if (!(di > 0)) continue;
                    const double tq = dist[i] / di;
                    if (best_i == 0x7fffffff || tq < best) { best = tq; best_i = i; }
                }
                wave_argmin(best, best_i);
                const int leaving = uni(best_i);
                if (leaving == 0x7fffffff) { st = SM_SINGULAR; break; }
                const double theta = uni(best);
                for (int i = lane; i < m; i += 64) {
                    const double tq = theta * d[i + 1];
                    const double v = dist[i] - tq;
                    dist[i] = v;
                    z[i] = v;
                }
                remove_entry(leaving);
                // the insertion again, on the reduced set
                common_with_new();
                sden = insertion_direction();
                if (uni(sden <= kSingularTol && sden >= -kSingularTol)) { st = SM_SINGULAR; break; }
                w_new = theta;
            } else if constexpr (SEQ) {
                if (m >= cap) { st = SM_SINGULAR; break; }   // a full set that rounding calls independent
                if (m + 2 > LD) { overflow = true; break; }  // next tier redoes this pass
            }
            const int cnt8 = round8(m + 1);
            const double invs = uni(1.0 / sden);
            // t1 = invs * d (zero beyond m)
#pragma unroll
            for (int q = 0; q * 64 < L::NA; ++q) {
                const int i = lane + 64 * q;
                if (i < L::NA) t1[i] = (i <= m) ? invs * d[i] : 0.0;
            }
            wsync();
            SM_T(5);
            // A[i][j] += (invs d_i) d_j : lane owns column j
            {   // (the column chunks of an 8-row batch together: one LDS round trip per batch)
                constexpr int NQA = (LD + 63) / 64;
                double dj[NQA];
                bool on[NQA];
#pragma unroll
                for (int q = 0; q < NQA; ++q) {
                    const int j = lane + 64 * q;
                    on[q] = j <= m;
                    dj[q] = on[q] ? d[j] : 0.0;
                }
                for (int i0 = 0; i0 < cnt8; i0 += 8) {
                    double a[NQA][8], t[8];
#pragma unroll
                    for (int q = 0; q < NQA; ++q)
                        if (q * 64 <= m && on[q]) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) a[q][u] = A[(i0 + u) * LD + lane + 64 * q];
                        }
                    ld8(t1 + i0, t);
#pragma unroll
                    for (int q = 0; q < NQA; ++q)
                        if (q * 64 <= m && on[q]) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const double pr = t[u] * dj[q];
                                A[(i0 + u) * LD + lane + 64 * q] = a[q][u] + pr;
                            }
                        }
                }
            }
            wsync();
            for (int i = lane; i <= m; i += 64) {
                const double v = -invs * d[i];
                A[i * LD + (m + 1)] = v;
                A[(m + 1) * LD + i] = v;
            }
            if (lane == 0) {
                A[(m + 1) * LD + (m + 1)] = invs;
                bsc[m + 1] = evalsum;
                if constexpr (SEQ) { z[m] = w_new; dist[m] = w_new; }
            }
            if (lane < WT) cfg[m * WT + lane] = newc[lane];
            ++m;
            changed = true;
            judged = false;
            wsync();
            SM_T(6);
        } else {
            // blocking constraints: argmin of dist/(dist - z) over {z < dist}, first index on ties
            double best = INFINITY;
            int best_i = 0x7fffffff;
            bool exist = false;
            for (int i = lane; i < m; i += 64) {
                const double zi = z[i], pi = dist[i];
                if (zi < pi && zi < 0) exist = true;
            }
            exist = __any(exist);
            if (exist) {   // (the ratios -- an IEEE division each -- only when a constraint blocks)
                for (int i = lane; i < m; i += 64) {
                    const double zi = z[i], pi = dist[i];
                    if (zi >= pi) continue;
                    const double tmp = pi / (pi - zi);
                    if (tmp < best) { best = tmp; best_i = i; }  // ascending i inside a lane
                }
            }
            if (!exist) {
                for (int i = lane; i < m; i += 64) dist[i] = z[i];
                changed = false;
                wsync();
                SM_T(8);
            } else {
                // wave argmin (value, then index)
                wave_argmin(best, best_i);
                double alpha = uni(best);
                const int blocking = uni(best_i);
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
                remove_entry(blocking);
                changed = true;
                judged = false;
                wsync();
                SM_T(9);
            }
        }
    }
    wsync();

    // outputs
    for (int i = lane; i < cap * W; i += 64) {
        const int row = i / W, w = i - row * W;
        aset_bits[s * cap * W + i] = (row < m) ? cfg[row * WT + w] : 0u;
    }
    for (int i = lane; i < cap; i += 64) p_out[s * cap + i] = (i < m) ? (float)dist[i] : 0.0f;
    if (a_out) {
        double* o = a_out + s * (int64_t)ldo * ldo;
        for (int i = 0; i <= m; ++i)
            for (int j = lane; j <= m; j += 64) o[i * ldo + j] = A[i * LD + j];
    }
    if (overflow) {
        double* vo = ws_vec + s * (int64_t)3 * ldo;
        for (int i = lane; i <= m; i += 64) {
            if (i < m) { vo[i] = z[i]; vo[ldo + i] = dist[i]; }
            vo[2 * ldo + i] = bsc[i];
        }
    }
    SM_TEND(s);
    if (lane == 0) {
        alen[s] = m;
        if (iters) iters[s] = it;
        if (status) status[s] = st;
        if (ws_state) {
            ws_state[s] = overflow ? (1 | ((int)polish << 1) | ((int)judged << 2) | (rebuilds << 4) | (rejudges << 8) | (last_rejudge << 12)) : 0;
            ws_it[s] = it;
            ws_tau[s] = tau;
        }
    }
}

// dist_jacobian_vec (lvmhelpers/sparsemap.py:38-44): tmp = S dp with S = invA[1:,1:], out = M tmp
// restricted to the first n (on) coordinates.
__global__ __launch_bounds__(64) void sparsemap_bwd_kernel(
    const float* __restrict__ dp, const uint32_t* __restrict__ aset_bits,
    const int32_t* __restrict__ alen, const double* __restrict__ inv_a, int64_t B, int n,
    float* __restrict__ dscores, float* __restrict__ dtrans)
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
    // SequenceSparseMAP.jv (sparsemap.py:79-87): d_eta_v[k] = sum over the configurations that
    // have bits k and k+1 on (sequence_binary.h:56-67)
    if (dtrans) {
        for (int k = lane; k + 1 < n; k += 64) {
            double acc = 0.0;
            for (int i = 0; i < m; ++i) {
                const uint32_t b0 = (cfg[i * W + (k >> 5)] >> (k & 31)) & 1u;
                const uint32_t b1 = (cfg[i * W + ((k + 1) >> 5)] >> ((k + 1) & 31)) & 1u;
                if (b0 & b1) acc += tmp[i];
            }
            dtrans[s * (n - 1) + k] = (float)acc;
        }
    }
}

}  // namespace smlvm

using namespace smlvm;

constexpr int kMaxDevices = 64;
static int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    return dev;
}
// compute units of the current device (per device: a process may drive several GPUs)
static int device_cus()
{
    static std::atomic<int> cus[kMaxDevices];
    const int dev = current_device();
    int c = cus[dev].load(std::memory_order_relaxed);
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0)
            c = 256;
        cus[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}

struct SmArgs {
    const float* scores; const float* trans; const double* init_scores; int32_t budget, max_iter; int64_t B; int32_t n;
    uint32_t* aset_bits; float* p; int32_t* alen; double* a_out; int32_t* iters; int32_t* status;
    int32_t* ws_state; int32_t* ws_it; double* ws_tau; double* ws_vec; hipStream_t stream;
};

template <int LD, int WT, int NK, bool SEQ>
static int launch_sm_fwd(const SmArgs& a, int resume)
{
    using L = SmLayout<LD, WT, NK>;
    static_assert(L::total <= 160 * 1024, "LDS per CU");
    // the attribute belongs to (kernel, device): remember it per device, not per thread
    static std::atomic<bool> configured[kMaxDevices];
    const int dev = current_device();
    if (L::total > 48 * 1024 && !configured[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)sparsemap_fwd_kernel<LD, WT, NK, SEQ>,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)L::total) != hipSuccess) {
            g_last_hip_error = (int)hipGetLastError();
            return SMLVM_ERR_HIP;
        }
        configured[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((sparsemap_fwd_kernel<LD, WT, NK, SEQ>), dim3((unsigned)a.B), dim3(64), L::total,
                       a.stream, a.scores, a.trans, a.init_scores, a.budget, a.max_iter, a.B, a.n,
                       a.aset_bits, a.p, a.alen, a.a_out, a.iters, a.status, resume, a.ws_state,
                       a.ws_it, a.ws_tau, a.ws_vec);
    return check_launch();
}

// Tier sizes for 64 < n <= 128 (LDS per sample 40.5 KB -> 4 per CU, 81.7 KB -> 2 per CU, 158 KB -> 1)
constexpr int kTier1 = 64, kTier2 = 94, kTier3 = SMLVM_SPARSEMAP_MAX_BITS + 2;
// Tier 0, only for batches that oversubscribe the chip at four samples per CU: 20 KB per sample ->
// eight per CU, i.e. two waves per SIMD, which hides part of the dependent-instruction latency
// that a lone wave per SIMD is exposed to; samples whose active set outgrows 38 entries move on.
constexpr int kTier0 = 40;
static_assert(SmLayout<kTier0, 4, 2>::total <= 20 * 1024, "tier 0: eight samples per CU");
static_assert(SmLayout<kTier1, 4, 2>::total <= 40 * 1024, "tier 1: four samples per CU");
static_assert(SmLayout<kTier2, 4, 2>::total <= 80 * 1024, "tier 2: two samples per CU");

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t smlvm_sparsemap_fwd_workspace_bytes(int64_t B, int32_t n, int32_t with_inv_a)
{
    if (B <= 0 || n <= 64 || n > SMLVM_SPARSEMAP_MAX_BITS) return 0;
    const size_t ldo = (size_t)n + 2;
    size_t bytes = align256(sizeof(int32_t) * 2 * (size_t)B) + align256(sizeof(double) * (size_t)B) +
                   align256(sizeof(double) * 3 * ldo * (size_t)B);
    if (!with_inv_a) bytes += align256(sizeof(double) * ldo * ldo * (size_t)B);
    return bytes;
}

// Shared dispatcher: SEQ = false the Bernoulli / budget factors, true the binary sequence factor.
template <bool SEQ>
static int run_sm_fwd(SmArgs a, void* workspace, size_t workspace_bytes)
{
    const int64_t B = a.B;
    const int n = a.n;
    if (n <= 8) return launch_sm_fwd<10, 1, 1, SEQ>(a, 0);
    if (n <= 16) return launch_sm_fwd<18, 1, 1, SEQ>(a, 0);
    if (n <= 32) return launch_sm_fwd<34, 1, 1, SEQ>(a, 0);
    if (n <= 64) return launch_sm_fwd<66, 2, 1, SEQ>(a, 0);
    // tiers
    const size_t need = smlvm_sparsemap_fwd_workspace_bytes(B, n, a.a_out != nullptr);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 7u))
        return SMLVM_ERR_WORKSPACE;
    const size_t ldo = (size_t)n + 2;
    char* w = static_cast<char*>(workspace);
    a.ws_state = reinterpret_cast<int32_t*>(w);
    a.ws_it = a.ws_state + B;
    w += align256(sizeof(int32_t) * 2 * (size_t)B);
    a.ws_tau = reinterpret_cast<double*>(w);
    w += align256(sizeof(double) * (size_t)B);
    a.ws_vec = reinterpret_cast<double*>(w);
    w += align256(sizeof(double) * 3 * ldo * (size_t)B);
    if (!a.a_out) a.a_out = reinterpret_cast<double*>(w);
    const int cus = device_cus();
    int rc;
    if (!SEQ && B > 4 * (int64_t)cus) {   // (the sequence factor, unused by the experiments, skips tier 0)
        rc = launch_sm_fwd<kTier0, 4, 2, false>(a, 0);
        if (rc != SMLVM_OK) return rc;
        rc = launch_sm_fwd<kTier1, 4, 2, SEQ>(a, 1);
    } else {
        rc = launch_sm_fwd<kTier1, 4, 2, SEQ>(a, 0);
    }
    if (rc != SMLVM_OK) return rc;
    if (n + 2 <= kTier2) return launch_sm_fwd<kTier2, 4, 2, SEQ>(a, 1);
    rc = launch_sm_fwd<kTier2, 4, 2, SEQ>(a, 1);
    if (rc != SMLVM_OK) return rc;
    return launch_sm_fwd<kTier3, 4, 2, SEQ>(a, 1);
}

extern "C" int smlvm_sparsemap_fwd(const float* scores, const double* init_scores, int32_t budget,
                                   int32_t max_iter, int64_t B, int32_t n, uint32_t* aset_bits,
                                   float* p, int32_t* alen, double* inv_a, int32_t* iters,
                                   int32_t* status, void* workspace, size_t workspace_bytes,
                                   void* stream)
{
    if (B < 0 || n < 1 || max_iter < 0 || budget < 0) return SMLVM_ERR_INVALID;
    if (n > SMLVM_SPARSEMAP_MAX_BITS) return SMLVM_ERR_UNSUPPORTED;
    if (B == 0) return SMLVM_OK;
    if (!scores || !aset_bits || !p || !alen) return SMLVM_ERR_INVALID;
    SmArgs a{scores, nullptr, init_scores, budget, max_iter, B, n, aset_bits, p, alen, inv_a, iters,
             status, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream};
    return run_sm_fwd<false>(a, workspace, workspace_bytes);
}

extern "C" int smlvm_sparsemap_seq_fwd(const float* scores, const float* trans,
                                       const double* init_scores, int32_t max_iter, int64_t B,
                                       int32_t n, uint32_t* aset_bits, float* p, int32_t* alen,
                                       double* inv_a, int32_t* iters, int32_t* status,
                                       void* workspace, size_t workspace_bytes, void* stream)
{
    if (B < 0 || n < 2 || max_iter < 0) return SMLVM_ERR_INVALID;
    if (n > SMLVM_SPARSEMAP_MAX_BITS) return SMLVM_ERR_UNSUPPORTED;
    if (B == 0) return SMLVM_OK;
    if (!scores || !trans || !aset_bits || !p || !alen) return SMLVM_ERR_INVALID;
    SmArgs a{scores, trans, init_scores, 0, max_iter, B, n, aset_bits, p, alen, inv_a, iters,
             status, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream};
    return run_sm_fwd<true>(a, workspace, workspace_bytes);
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
                       dp, aset_bits, alen, inv_a, B, n, dscores, (float*)nullptr);
    return check_launch();
}

extern "C" int smlvm_sparsemap_seq_bwd(const float* dp, const uint32_t* aset_bits,
                                       const int32_t* alen, const double* inv_a, int64_t B,
                                       int32_t n, float* dscores, float* dtrans, void* stream)
{
    if (B < 0 || n < 2) return SMLVM_ERR_INVALID;
    if (n > SMLVM_SPARSEMAP_MAX_BITS) return SMLVM_ERR_UNSUPPORTED;
    if (B == 0) return SMLVM_OK;
    if (!dp || !aset_bits || !alen || !inv_a || !dscores || !dtrans) return SMLVM_ERR_INVALID;
    const int cap = n + 1, W = (n + 31) / 32;
    const size_t lds = sizeof(double) * 2 * cap + sizeof(uint32_t) * (size_t)cap * W;
    hipLaunchKernelGGL(sparsemap_bwd_kernel, dim3((unsigned)B), dim3(64), lds, (hipStream_t)stream,
                       dp, aset_bits, alen, inv_a, B, n, dscores, dtrans);
    return check_launch();
}
