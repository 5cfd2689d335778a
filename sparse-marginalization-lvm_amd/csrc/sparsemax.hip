// Kernel 1: batched row-wise sparsemax forward / backward for gfx950 (wave64).
//
// Replaces entmax.sparsemax as called at lvmhelpers/marg.py:45,51 and lvmhelpers/structmarg.py:48.
//
// Design (HBM-bound: 8K B/row forward, 12K B/row backward, one read and one write per element):
//   * a row lives entirely in registers: G lanes x E elements (G = 8..64 lanes of one wave for
//     K <= 4096, several rows packed per wave when K is small; a whole workgroup of NW waves for
//     4096 < K <= 65536), loaded with one coalesced 16-byte load per lane per 4 elements;
//   * the threshold tau is found without sorting: Michelot's fixed point iteration (a Newton
//     iteration from the left on the piecewise-linear equation sum(max(z - tau, 0)) = 1) started
//     from the bound tau >= max - 1, each step one pass over the registers plus a DPP all-reduce
//     inside the lane group;
//   * the result is then made bit-identical to the sort-based definition (oracle/sparsemax_oracle.c:
//     fp64-accumulated prefix sums rounded to fp32, fp32 test j*z_(j) > c_j): the boundary ranks
//     are re-tested with exactly those fp32 expressions, from an fp64 sum of the candidate set,
//     and the set is moved rank by rank until the test flips exactly at its edge;
//   * epilogue fused: p = max(z - tau, 0), support size, #{p>0}, optional entropy.
// Backward: dx = [p != 0] * (dy - sum(dy[p != 0]) / supp_size), fp64 row sum; rows with a small
// support gather dy at the support instead of streaming the row (12K -> ~8K bytes of traffic).
#include "common.hpp"

#include <float.h>
#include <math.h>

namespace smlvm {

// Wave number inside the workgroup as a scalar: LDS addresses derived from it stay out of the VGPRs.
__device__ __forceinline__ int wave_index() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
// A float every lane computes identically (from kernel arguments), parked in a scalar register.
__device__ __forceinline__ float uniform(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// ------------------------------------------------------------------------------- reducers
// A reducer combines per-lane partials over everything that holds one row.  `cs` and `edge`
// reduce several values in ONE step: for the workgroup reducer that is one barrier, and it keeps
// the per-element compare masks from having to stay alive (in SGPRs) from one pass to the next.
template <int G>
struct WaveRed {
    static constexpr int kLanes = G;
    // position of this lane among the lanes that hold the row, and how many there are
    __device__ __forceinline__ int idx() const { return (int)(threadIdx.x & 63) % G; }
    __device__ __forceinline__ int stride() const { return G; }
    __device__ __forceinline__ float max(float v) { return group_reduce<G>(v, OpMaxF{}); }
    __device__ __forceinline__ float min(float v) { return group_reduce<G>(v, OpMinF{}); }
    __device__ __forceinline__ int sum(int v) { return group_reduce<G>(v, OpSum{}); }
    __device__ __forceinline__ double sum(double v) { return group_reduce<G>(v, OpSum{}); }
    __device__ __forceinline__ void cs(int& c, float& s)
    {
        c = group_reduce<G>(c, OpSum{});
        s = group_reduce<G>(s, OpSum{});
    }
    __device__ __forceinline__ void edge(int& mg, int& t, double& S, float& vn)
    {
        mg = group_reduce<G>(mg, OpSum{});
        t = group_reduce<G>(t, OpSum{});
        S = group_reduce<G>(S, OpSum{});
        vn = group_reduce<G>(vn, OpMaxF{});
    }
    __device__ __forceinline__ void sum2(double& a, double& b)
    {
        a = group_reduce<G>(a, OpSum{});
        b = group_reduce<G>(b, OpSum{});
    }
    // two sums and a count (entmax15's closed form)
    __device__ __forceinline__ void sums(double& a, double& b, int& c)
    {
        a = group_reduce<G>(a, OpSum{});
        b = group_reduce<G>(b, OpSum{});
        c = group_reduce<G>(c, OpSum{});
    }
    // count, sum, smallest of a set and the largest element outside it
    __device__ __forceinline__ void first(int& m, double& S, float& lo, float& vn)
    {
        m = group_reduce<G>(m, OpSum{});
        S = group_reduce<G>(S, OpSum{});
        lo = group_reduce<G>(lo, OpMinF{});
        vn = group_reduce<G>(vn, OpMaxF{});
    }
};

// Whole-workgroup all-reduce through LDS, one barrier per call: each wave reduces with DPP, lane 0
// parks the wave totals in an LDS record, and after the barrier every wave re-reduces the NW
// records (one per lane, identity elsewhere) with the same DPP tree, so all threads get the same
// bits.  Three rotating record sets: a wave can be at most one reduction ahead of the slowest one,
// whose reads of the previous set are done before it reaches the barrier that lets anyone write
// the set after next.
template <int NW>
struct BlockRed {
    static_assert(NW <= 16, "second level is one 16-lane DPP row");
    static constexpr int kSlots = 3;            // 8-byte slots per wave record
    static constexpr int kLdsWords = 3 * NW * kSlots;  // unsigned long long words
    unsigned long long* base;
    int phase;
    __device__ __forceinline__ BlockRed(unsigned long long* lds) : base(lds), phase(0) {}
    __device__ __forceinline__ int idx() const { return (int)threadIdx.x; }
    __device__ __forceinline__ int stride() const { return 64 * NW; }
    // slot index of lane's record, recomputed where it is used (kept alive across the row loop it
    // would be spilled, and every reduction would start with a scratch reload)
    static __device__ __forceinline__ int rec_of(int lane)
    {
        asm volatile("" : "+v"(lane));
        return lane * kSlots;
    }
    __device__ __forceinline__ unsigned long long* next_set()
    {
        unsigned long long* s = base + phase * NW * kSlots;
        phase = (phase == 2) ? 0 : phase + 1;
        return s;
    }
    template <class T>
    static __device__ __forceinline__ unsigned long long pack(T v)
    {
        unsigned long long bits = 0;
        __builtin_memcpy(&bits, &v, sizeof(T));
        return bits;
    }
    template <class T>
    static __device__ __forceinline__ T unpack(unsigned long long bits)
    {
        T v;
        __builtin_memcpy(&v, &bits, sizeof(T));
        return v;
    }
    template <class T>
    static __device__ __forceinline__ T bcast0(T v)
    {
        if constexpr (sizeof(T) == 8)
            return __builtin_bit_cast(T, readlane_f64(__builtin_bit_cast(double, v), 0));
        else
            return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    }
    template <class T, class Op>
    __device__ __forceinline__ T all(T v, Op op, T identity)
    {
        v = group_reduce<64>(v, op);
        unsigned long long* s = next_set();
        const int lane = threadIdx.x & 63;
        if (lane == 0) s[wave_index() * kSlots] = pack(v);
        __syncthreads();
        T part = identity;
        if (lane < NW) part = unpack<T>(s[rec_of(lane)]);
        return bcast0(group_reduce<16>(part, op));
    }
    __device__ __forceinline__ float max(float v) { return all(v, OpMaxF{}, -INFINITY); }
    __device__ __forceinline__ float min(float v) { return all(v, OpMinF{}, INFINITY); }
    __device__ __forceinline__ int sum(int v) { return all(v, OpSum{}, 0); }
    __device__ __forceinline__ double sum(double v) { return all(v, OpSum{}, 0.0); }
    __device__ __forceinline__ void cs(int& c, float& sm)
    {
        c = group_reduce<64>(c, OpSum{});
        sm = group_reduce<64>(sm, OpSum{});
        unsigned long long* s = next_set();
        const int lane = threadIdx.x & 63;
        if (lane == 0) {
            unsigned long long* rec = s + wave_index() * kSlots;
            rec[0] = pack(c);
            rec[1] = pack(sm);
        }
        __syncthreads();
        int pc = 0;
        float ps = 0.0f;
        if (lane < NW) {
            const int r = rec_of(lane);
            pc = unpack<int>(s[r]);
            ps = unpack<float>(s[r + 1]);
        }
        c = bcast0(group_reduce<16>(pc, OpSum{}));
        sm = bcast0(group_reduce<16>(ps, OpSum{}));
    }
    __device__ __forceinline__ void edge(int& mg, int& t, double& S, float& vn)
    {
        mg = group_reduce<64>(mg, OpSum{});
        t = group_reduce<64>(t, OpSum{});
        S = group_reduce<64>(S, OpSum{});
        vn = group_reduce<64>(vn, OpMaxF{});
        unsigned long long* s = next_set();
        const int lane = threadIdx.x & 63;
        if (lane == 0) {
            unsigned long long* rec = s + wave_index() * kSlots;
            rec[0] = ((unsigned long long)(unsigned)t << 32) | (unsigned)mg;
            rec[1] = pack(S);
            rec[2] = pack(vn);
        }
        __syncthreads();
        int pm = 0, pt = 0;
        double pS = 0.0;
        float pv = -INFINITY;
        if (lane < NW) {
            const int r = rec_of(lane);
            unsigned long long w = s[r];
            pm = (int)(unsigned)(w & 0xffffffffull);
            pt = (int)(unsigned)(w >> 32);
            pS = unpack<double>(s[r + 1]);
            pv = unpack<float>(s[r + 2]);
        }
        mg = bcast0(group_reduce<16>(pm, OpSum{}));
        t = bcast0(group_reduce<16>(pt, OpSum{}));
        S = bcast0(group_reduce<16>(pS, OpSum{}));
        vn = bcast0(group_reduce<16>(pv, OpMaxF{}));
    }
    __device__ __forceinline__ void sum2(double& a, double& b)   // one barrier for both
    {
        a = group_reduce<64>(a, OpSum{});
        b = group_reduce<64>(b, OpSum{});
        unsigned long long* s = next_set();
        const int lane = threadIdx.x & 63;
        if (lane == 0) {
            unsigned long long* rec = s + wave_index() * kSlots;
            rec[0] = pack(a);
            rec[1] = pack(b);
        }
        __syncthreads();
        double pa = 0.0, pb = 0.0;
        if (lane < NW) {
            const int r = rec_of(lane);
            pa = unpack<double>(s[r]);
            pb = unpack<double>(s[r + 1]);
        }
        a = bcast0(group_reduce<16>(pa, OpSum{}));
        b = bcast0(group_reduce<16>(pb, OpSum{}));
    }
    __device__ __forceinline__ void sums(double& a, double& b, int& c)   // one barrier for the three
    {
        a = group_reduce<64>(a, OpSum{});
        b = group_reduce<64>(b, OpSum{});
        c = group_reduce<64>(c, OpSum{});
        unsigned long long* s = next_set();
        const int lane = threadIdx.x & 63;
        if (lane == 0) {
            unsigned long long* rec = s + wave_index() * kSlots;
            rec[0] = pack(a);
            rec[1] = pack(b);
            rec[2] = pack(c);
        }
        __syncthreads();
        double pa = 0.0, pb = 0.0;
        int pc = 0;
        if (lane < NW) {
            const int r = rec_of(lane);
            pa = unpack<double>(s[r]);
            pb = unpack<double>(s[r + 1]);
            pc = unpack<int>(s[r + 2]);
        }
        a = bcast0(group_reduce<16>(pa, OpSum{}));
        b = bcast0(group_reduce<16>(pb, OpSum{}));
        c = bcast0(group_reduce<16>(pc, OpSum{}));
    }
    __device__ __forceinline__ void first(int& m, double& S, float& lo, float& vn)
    {
        m = group_reduce<64>(m, OpSum{});
        S = group_reduce<64>(S, OpSum{});
        lo = group_reduce<64>(lo, OpMinF{});
        vn = group_reduce<64>(vn, OpMaxF{});
        unsigned long long* s = next_set();
        const int lane = threadIdx.x & 63;
        if (lane == 0) {
            unsigned long long* rec = s + wave_index() * kSlots;
            rec[0] = ((unsigned long long)__builtin_bit_cast(unsigned, lo) << 32) | (unsigned)m;
            rec[1] = pack(S);
            rec[2] = pack(vn);
        }
        __syncthreads();
        int pm = 0;
        double pS = 0.0;
        float pl = INFINITY, pv = -INFINITY;
        if (lane < NW) {
            const int r = rec_of(lane);
            unsigned long long w = s[r];
            pm = (int)(unsigned)(w & 0xffffffffull);
            pl = __builtin_bit_cast(float, (unsigned)(w >> 32));
            pS = unpack<double>(s[r + 1]);
            pv = unpack<float>(s[r + 2]);
        }
        m = bcast0(group_reduce<16>(pm, OpSum{}));
        S = bcast0(group_reduce<16>(pS, OpSum{}));
        lo = bcast0(group_reduce<16>(pl, OpMinF{}));
        vn = bcast0(group_reduce<16>(pv, OpMaxF{}));
    }
};

// ------------------------------------------------------------------------------- exact edge
// The oracle's support size is the NUMBER of ranks j = 1..K whose fp32 test
//     (float)j * z_(j)  >  (float)S_j - 1.0f          (S_j: fp64 prefix sum of the sorted row)
// holds, and tau = ((float)S_s - 1)/s with S_s the prefix sum at that count.  In exact arithmetic
// f(j) = 1 + j z_(j) - S_j is non-increasing, constant inside a group of tied values, positive on
// the support and negative below it; the fp32 evaluation is off by at most
//     noise(j) <= 2^-24 (j|z_j| + 2|S_j| + 1) <= 2^-24 (3 j |z_j| + 1)      (all z <= 0)
// so the outcome at a rank is only open when |f(j)| is inside that noise.  The search therefore
// works on tie groups (value v, t elements, m elements above with fp64 sum S; f = 1 + m v - S for
// every rank of the group):
//   * f >  B(m, t, v) = 2^-23 (3 (m + t)|v| + 1): the group and everything above pass   (robust)
//   * f < -B: the group fails;   f(next group) < -2^-23 (3 N |v_next| + 1), N = elements the
//     search can see: every deeper rank fails                                            (robust)
//   * otherwise the group's t ranks are evaluated one by one with exactly the oracle's fp32
//     expressions (spread over the lanes that hold the row) and the passing ones counted.
// The walk goes up from Michelot's fixed point to the first robust-pass group, then down through
// the undecided groups; the count of passing ranks and the prefix sum at that count are the
// oracle's s and S_s bit for bit -- including rows where hundreds or thousands of elements tie at
// or within an ulp of the threshold and the oracle's test passes at isolated ranks.

// The oracle's test at rank j = m_gt + r inside the tie group of value v (r = 1..t), given the
// count m_gt and fp64 sum S_gt of the elements strictly above v:
//     (float)j * v  >  (float)(S_gt + r*v) - 1.0f
__device__ __forceinline__ bool cond_rank(int m_gt, double S_gt, float v, int r)
{
    float j = (float)(m_gt + r);
    double S = S_gt + (double)r * (double)v;  // r*v is exact in fp64
    float c = (float)S - 1.0f;
    return j * v > c;
}

// +1: every rank of the group (and of every group above) passes; -1: every rank of the group
// fails; 0: inside the rounding noise of the fp32 test, count rank by rank
__device__ __forceinline__ int classify_group(int m_gt, double S_gt, float v, int t)
{
    const double d = (double)m_gt * (double)v - S_gt + 1.0;
    const double B = 0x1p-23 * (3.0 * (double)(m_gt + t) * fabs((double)v) + 1.0);
    return d > B ? 1 : (d < -B ? -1 : 0);
}

// Exponent of the lowest set bit of a finite, non-zero double (x is an odd multiple of 2^that).
__device__ __forceinline__ int lsb_exp_f64(double x)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    int e = (int)((b >> 52) & 0x7ffull);
    unsigned long long m = b & ((1ull << 52) - 1ull);
    if (e != 0) m |= 1ull << 52; else e = 1;
    return e - 1075 + __builtin_ctzll(m);
}

// True when every fp32 / fp64 operation of cond_rank(m_gt, S_gt, v, r), r = 1..t, is EXACT: v, S_gt
// and 1 are multiples of one power of two 2^q (q <= 0) and every intermediate -- (float)j, j v,
// S_gt + r v, (float)S, (float)S - 1 -- is a multiple of 2^q below 2^(24+q) in magnitude, hence
// representable in fp32.  Then the test at every rank of the group is the exact inequality
// (m_gt + r) v > S_gt + r v - 1, i.e. m_gt v - S_gt + 1 > 0, the same for all r: the group passes
// or fails as a whole and nothing has to be counted rank by rank (round 3: integer- and
// dyadic-valued rows -- the "exact tie at the threshold" rows of SURVEY.md 8(d) S1 -- used to run
// up to 65535 rank tests per row).
__device__ __forceinline__ bool tie_group_is_exact(int m_gt, double S_gt, float v, int t)
{
    int q = 0;
    if (v != 0.0f) { const int e = lsb_exp_f64((double)v); q = e < q ? e : q; }
    if (S_gt != 0.0) { const int e = lsb_exp_f64(S_gt); q = e < q ? e : q; }
    if (q < -900 || m_gt + t >= (1 << 24)) return false;
    const double lim = __builtin_bit_cast(double, (unsigned long long)(1023 + 24 + q) << 52);   // 2^(24+q)
    const double av = fabs((double)v);
    return (double)(m_gt + t) * av < lim && fabs(S_gt) + (double)t * av + 1.0 < lim;
}
// number of ranks of an undecided group whose test holds, when tie_group_is_exact
__device__ __forceinline__ int exact_tie_count(int m_gt, double S_gt, float v, int t)
{
    return ((double)m_gt * (double)v - S_gt + 1.0 > 0.0) ? t : 0;
}

// every rank below (m, S), whose largest value is vnext, fails robustly (nvis >= ranks in sight)
__device__ __forceinline__ bool deeper_all_fail(int m, double S_m, float vnext, int nvis)
{
    const double d = (double)m * (double)vnext - S_m + 1.0;
    return d < -0x1p-23 * (3.0 * (double)nvis * fabs((double)vnext) + 1.0);
}

// This thread's share of the ranks r = 1..t of a tie group whose test holds (r = first + 1,
// first + 1 + stride, ...).  Rare (ties inside the rounding noise of the edge), so it is kept out
// of line: inlined it cost the small-row kernels their register budget.
__device__ __noinline__ int count_ranks_part(int m_gt, double S_gt, float v, int t, int first,
                                             int stride)
{
    int n = 0;
    for (int r = first + 1; r <= t; r += stride) n += cond_rank(m_gt, S_gt, v, r) ? 1 : 0;
    return n;
}

// Elements at or below this value (max-shifted frame) can neither be in the support (tau >= -1)
// nor pass the oracle's fp32 test by rounding: for z_j = -u, f(j) <= s (|tau| - u) <= 1 - u, and
// 1 - u < -2^-24 (3 K u + 1) once u > (1 + 2^-23) / (1 - 3 K 2^-23).  Candidate lists keep what is
// above it (K <= 2^20; wider rows run the search on the whole row).
constexpr int kCandMaxK = 1 << 20;
__device__ __forceinline__ float cand_floor(int K)
{
    const float a = 3.0f * (float)K * 0x1p-23f;
    return uniform(-((1.0f + 0x1p-23f) / (1.0f - a)) * (1.0f + 1e-5f));
}

struct Edge {
    int m_gt, t;
    double S_gt;
    float vnext;
};

// A source hands every element of the row (max-shifted) to f, in a fixed per-thread order.
template <int E>
struct RegSrc {
    const float (&z)[E];
    template <class F>
    __device__ __forceinline__ void each(F&& f) const
    {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            float v = z[e];
            // opaque copy: stops LICM from hoisting 64 float->double conversions (128 VGPRs) out of
            // the edge loop, and the compare masks of one pass from staying alive into the next
            asm volatile("" : "+v"(v));
            f(v);
        }
    }
};

template <class Src, class Red>
__device__ __forceinline__ Edge edge_pass(const Src& src, float lo, Red& red)
{
    int mg = 0, t = 0;
    double S = 0.0;
    float vn = -INFINITY;
    src.each([&](float v) {
        const bool gt = v > lo;
        mg += gt ? 1 : 0;
        t += (v == lo) ? 1 : 0;
        S += gt ? (double)v : 0.0;
        vn = (v < lo) ? fmaxf(vn, v) : vn;
    });
    red.edge(mg, t, S, vn);
    Edge r;
    r.m_gt = mg;
    r.t = t;
    r.S_gt = S;
    r.vnext = vn;
    return r;
}

// The exact edge from a lower bound tau of the threshold that is at (or within rounding of)
// Michelot's fixed point.  nvis: upper bound of the number of elements src shows.  Returns the
// oracle's support size s and the fp64 sum S of the s largest elements.  Every decision below comes
// out of `red`, so control flow is uniform over the lanes that hold the row.
template <class Src, class Red>
__device__ __forceinline__ void solve_edge(const Src& src, Red& red, float tau, int nvis, int& s_out,
                                           double& S_out, int* m_robust = nullptr)
{
    // One pass at tau: the set {v > tau} (count m_all, fp64 sum S_all), its smallest value lo and
    // the largest value outside it.  For the tie group at lo, f = 1 + m_gt lo - S_gt is the same
    // number as 1 + m_all lo - S_all (the group's own t lo cancels), so the usual case -- that
    // group passes robustly -- is decided without counting the group.
    int m_top = 0;
    double S_top = 0.0;
    float lo = INFINITY, vnext = -INFINITY;
    src.each([&](float v) {
        const bool gt = v > tau;
        m_top += gt ? 1 : 0;
        S_top += gt ? (double)v : 0.0;
        lo = gt ? fminf(lo, v) : lo;
        vnext = gt ? vnext : fmaxf(vnext, v);
    });
    red.first(m_top, S_top, lo, vnext);
    bool top_ok = false;
    if (m_top > 0) {
        const double d = (double)m_top * (double)lo - S_top + 1.0;
        top_ok = d > 0x1p-23 * (3.0 * (double)m_top * fabs((double)lo) + 1.0);   // classify_group's bound
    }
    if (__builtin_expect(!top_ok, 0)) {
        // up to the first group that passes robustly (the top group always does: f = 1)
        Edge b;
        for (;;) {
            b = edge_pass(src, lo, red);
            if (b.m_gt == 0 || classify_group(b.m_gt, b.S_gt, lo, b.t) > 0) break;
            float up = INFINITY;
            src.each([&](float v) { up = (v > lo) ? fminf(up, v) : up; });
            lo = red.min(up);
        }
        m_top = b.m_gt + b.t;
        S_top = b.S_gt + (double)b.t * (double)lo;
        vnext = b.vnext;
    }
    if (m_robust) *m_robust = m_top;   // these ranks are in the support in exact arithmetic too

    // down through the groups the noise bound cannot decide
    int m = m_top, C = 0;
    double S = S_top, S_at = S_top;
    const float first_lo = vnext;
    bool open = true, dirty = false;   // open: the passing ranks so far are a prefix of the ranks
    while (vnext > -INFINITY) {
        if (__builtin_expect(deeper_all_fail(m, S, vnext, nvis), 1)) break;
        const Edge g = edge_pass(src, vnext, red);
        const int k = classify_group(g.m_gt, g.S_gt, vnext, g.t);
        int cnt = k > 0 ? g.t : 0;
        if (__builtin_expect(k == 0, 0)) {
            if (tie_group_is_exact(g.m_gt, g.S_gt, vnext, g.t)) cnt = exact_tie_count(g.m_gt, g.S_gt, vnext, g.t);
            else cnt = red.sum(count_ranks_part(g.m_gt, g.S_gt, vnext, g.t, red.idx(), red.stride()));
        }
        if (open) {
            S_at = g.S_gt + (double)cnt * (double)vnext;
            open = cnt == g.t;
        } else if (cnt > 0) {
            dirty = true;
        }
        C += cnt;
        m = g.m_gt + g.t;
        S = g.S_gt + (double)g.t * (double)vnext;
        vnext = g.vnext;
    }
    int s = m_top + C;
    if (__builtin_expect(dirty, 0)) {
        // ranks passed below a rank that failed: the prefix sum at the count s is read off a second
        // walk over the same groups (needs two undecided groups in a row: practically never)
        int rem = C;
        float lo2 = first_lo;
        for (;;) {
            const Edge g = edge_pass(src, lo2, red);
            const int take = rem < g.t ? rem : g.t;
            S_at = g.S_gt + (double)take * (double)lo2;
            rem -= take;
            if (rem <= 0 || !(g.vnext > -INFINITY)) break;
            lo2 = g.vnext;
        }
    }
    if (s < 1) { s = 1; S_at = 0.0; }   // rows without a finite maximum (inf / NaN input)
    s_out = s;
    S_out = S_at;
}

// Threshold search on max-shifted values v[] (padding = -inf; every element above cand_floor(K)
// must be present): Michelot iteration, then the exact edge.
template <int E, class Red>
__device__ __forceinline__ void solve_core(const float (&z)[E], Red& red, int nvis, int& s_out,
                                           double& S_out, int* m_robust = nullptr, float tau0 = -1.0f,
                                           bool at_fixed_point = false)
{
    // --- Michelot iteration from a lower bound of the threshold (tau >= -1: z_max = 0 after the
    // shift; callers may know a better one).  A start that fp32 rounding put a hair above the true
    // threshold is as good: the edge search below walks down from where the iteration stops.
    float tau = tau0;
    int cnt_prev = 0x7fffffff;
    while (!at_fixed_point) {   // (uniform over everything `red` spans: the caller ran the iteration)
        int c = 0;
        float s = 0.0f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            bool in = z[e] > tau;
            c += in ? 1 : 0;
            s += in ? z[e] : 0.0f;
        }
        red.cs(c, s);
        if (c >= cnt_prev) break;  // the set stopped shrinking: fixed point (up to fp32 rounding)
        cnt_prev = c;
        tau = fmaxf(tau, (s - 1.0f) / (float)c);
    }
    // (tau is laundered so the compiler recomputes these compares instead of keeping the last
    // iteration's 64 lane masks alive in SGPRs across the reductions)
    asm volatile("" : "+v"(tau));
    const RegSrc<E> src{z};
    solve_edge(src, red, tau, nvis, s_out, S_out, m_robust);
}

// ------------------------------------------------------------------------------- candidates
// Only elements within 1 of the row maximum can be in the support (tau >= max - 1).  With ML-like
// logits that is a few dozen of K, so the threshold search runs on a compacted candidate list
// (through LDS, <= kCandPerLane per lane) instead of on all E registers; rows with more candidates
// than the list holds (near-uniform rows) take the full-width search.
constexpr int kCandPerLane = 4;
constexpr int kCandCap = 64 * kCandPerLane;  // candidates per row on the fast path

__device__ __forceinline__ int lanes_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// ------------------------------------------------------------------------------- support lists
// Optional by-product of the forward: the columns of the non-zeros of every row, so that the
// backward does not have to read p at all (p is zero almost everywhere; the backward then only
// gathers dy at the listed columns and zero-fills dx).  Layout [R, cap] int32, cap <= 64: the
// columns of p > 0 in no particular order, -1 in the unused slots; a first entry of -2 means "no
// list for this row" (more than cap non-zeros, or a row wider than one wave's tile) and sends the
// backward to the p-based path.
constexpr int kColsNone = -1, kColsOverflow = -2, kColsMask = -3, kColsMaxCap = 64;
// rows wider than a wave's tile (K > 4096: 16 KB and more per row) may carry longer lists: the
// segment backward reads two slots per lane
constexpr int kColsMaxCapWide = 128;
// kColsMask in slot 0: no list, but the row's non-zero pattern is in the support-mask buffer (one
// 64-bit word per lane of the <64, E> tile, bit e = register e: a layout private to this file and
// only written / read by the 16-byte-layout kernels with E >= 32, i.e. 1024 < K <= 4096).  The
// backward then gathers dy at the set bits without reading p.
constexpr int kMaskWordsPerRow = 128;   // 64 lanes x 2 x uint32

template <int E, bool VEC>
__device__ __forceinline__ int col_of(int e, int gl, int GT)
{
    return VEC ? (((e >> 2) * GT + gl) * 4 + (e & 3)) : (e * GT + gl);
}

// bits of a wave-wide ballot that belong to this lane's group of G lanes: (count, count below me)
template <int G>
__device__ __forceinline__ void group_count(unsigned long long m, int gl, int grp, int& cnt, int& below)
{
    if constexpr (G == 64) {
        cnt = __popcll(m);
        below = lanes_below(m);
    } else {
        const unsigned group_bits = (G == 32) ? 0xffffffffu : ((1u << G) - 1u);
        const unsigned mine = (unsigned)(m >> (grp * G)) & group_bits;
        cnt = __popc(mine);
        below = __popc(mine & ((1u << gl) - 1u));
    }
}


// inclusive prefix sum over the G consecutive lanes of a group (gl = lane % G)
template <int G>
__device__ __forceinline__ int group_scan_incl(int v, int gl)
{
    static_assert(G == 8 || G == 16 || G == 32, "lane groups inside one wave");
#define SMLVM_SCAN_STEP(D)                                                                       \
    if constexpr (G > D) {                                                                       \
        const int o = __builtin_amdgcn_update_dpp(0, v, 0x110 + D, 0xF, 0xF, true); /* row_shr:D */ \
        v += (G == 16 || gl >= D) ? o : 0;   /* 16-lane rows: the shift itself fills with zero */  \
    }
    SMLVM_SCAN_STEP(1)
    SMLVM_SCAN_STEP(2)
    SMLVM_SCAN_STEP(4)
    SMLVM_SCAN_STEP(8)
#undef SMLVM_SCAN_STEP
    if constexpr (G == 32) {   // second 16-lane row: add the first row's total (its lane 15)
        const int o = __shfl(v, ((int)(threadIdx.x & 63) & ~31) | 15);
        v += gl >= 16 ? o : 0;
    }
    return v;
}

// Appends this wave's elements > thr to list[0..cap) in lane order per register; returns how many
// there are (may exceed cap: then the list is incomplete and the caller must not use it).
// colbase/colstep: column of register e of this lane = colbase + (e >> 2) * colstep + (e & 3) for
// the 16-byte layout, colbase + e * colstep for the scalar one (recorded next to the values when
// `cols` is not null, for the support lists).
template <int E, bool VEC>
__device__ __forceinline__ int compact_candidates(const float (&z)[E], float thr, float* list,
                                                  int cap, int* cols, int colbase, int colstep)
{
    int base = 0;
    // opaque copy: the E column numbers derived from it must not be hoisted out of the caller's
    // retry loop (64 registers, spilled, reloaded per element); they cost one add where they are used
    asm volatile("" : "+v"(colbase));
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool in = z[e] > thr;
        const unsigned long long m = __ballot(in);
        if (m != 0ull && base <= cap) {  // wave-uniform; past cap the list is already useless
            const int off = base + lanes_below(m);
            if (in && off < cap) {
                list[off] = z[e];
                if (cols) cols[off] = colbase + (VEC ? (e >> 2) * colstep + (e & 3) : e * colstep);
            }
            base += __popcll(m);
        }
    }
    return base;
}

// Rows with more than kCandCap elements above the floor (narrow logit distributions): a few
// full-width Michelot steps raise the bound tau_k <= tau first, and the candidates are taken above
// tau_k - kLazyMargin instead.  The fp32 sums behind tau_k are off by far less than the margin
// (<= 70 roundings of 2^-24 each on values in [-1, 0]), so the list holds the whole support.  That
// it also holds every element the oracle's fp32 test could admit by rounding is checked AFTER the
// search on the list (lazy_list_is_complete): with m ranks known to be in the support and
// tau >= tau_k, an element at -u has f <= -m (u - |tau_k|), which is below the test's noise
// 2^-23 (3 K u + 1) for every u >= |thr| once |thr| (m - 3 K 2^-23) > m |tau_k| + 2^-23.  A row
// that fails the check (a short support under a very wide row) runs the full-width search.
constexpr int kLazyPasses = 8;
constexpr int kLazyTry = 2 * kCandCap;      // attempt the compaction once |{z > tau_k}| is below this
constexpr float kLazyMargin = 1e-3f;
__device__ __forceinline__ bool lazy_list_is_complete(float thr, float tau_k, int m_robust, int K)
{
    const float a = uniform(3.0f * (float)K * 0x1p-23f);
    const float m = (float)m_robust;
    const float tl = fabsf(tau_k) + 1e-5f;   // |lower bound of tau|, fp32 slop of tau_k included
    return m > a && fabsf(thr) * (m - a) > (m * tl + 0x1p-23f) * (1.0f + 1e-5f);
}
// the highest cut that passes the check above (-inf when none does: use the floor)
__device__ __forceinline__ float lazy_complete_thr(float tau_k, int m_robust, int K)
{
    const float a = uniform(3.0f * (float)K * 0x1p-23f);
    const float m = (float)m_robust;
    const float tl = fabsf(tau_k) + 1e-5f;
    return m > a ? -((m * tl + 0x1p-23f) * (1.0f + 1e-5f) / (m - a)) * (1.0f + 1e-5f) - 1e-6f : -INFINITY;
}

// The threshold of any SUBSET of a row is a lower bound of the row's threshold (more elements can
// only raise tau).  One value per lane -- the lane's own maximum -- through a wave-local Michelot
// iteration gives a bound that is already close for narrow distributions, at ~30 instructions a
// step and no pass over the registers; it replaces the first full-width steps above.
__device__ __forceinline__ float subset_bound(float v)
{
    WaveRed<64> wred;
    float tau = -1.0f;
    int cnt_prev = 0x7fffffff;
    for (;;) {
        const bool in = v > tau;
        int c = in ? 1 : 0;
        float sm = in ? v : 0.0f;
        wred.cs(c, sm);
        if (c >= cnt_prev) break;
        cnt_prev = c;
        tau = fmaxf(tau, (sm - 1.0f) / (float)c);
    }
    return tau;
}

// Support list of a row from Q candidates per lane (value c[q], column cc[q]): the non-zeros of p
// are the candidates above tau.  One lane group of G lanes per row.  The lane that fills slot 0
// is also the one that overwrites it with the overflow mark (same thread, program order).
template <int Q, int G>
__device__ __forceinline__ int emit_cols_from_candidates(const float (&c)[Q], const int (&cc)[Q],
                                                          float tau, int gl, int grp,
                                                          int32_t* __restrict__ list, int cap, bool live,
                                                          int mark = kColsOverflow, long long vdelta = 0)
{
    // vdelta != 0: the row's probabilities go to the value list at byte distance vdelta from the
    // column list, slot for slot (smlvm_sparsemax_fwd_lists: the compaction then needs no gather)
    int base = 0;
    bool wrote0 = false;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const float pv = c[q] - tau;
        const bool nz = pv > 0.0f;
        const unsigned long long m = __ballot(nz);
        if (m != 0ull) {
            int cnt, below;
            group_count<G>(m, gl, grp, cnt, below);
            const int off = base + below;
            if (nz && live && off < cap) {
                list[off] = cc[q];
                wrote0 = wrote0 || off == 0;
                if (vdelta) *reinterpret_cast<float*>(reinterpret_cast<char*>(list + off) + vdelta) = pv;
            }
            base += cnt;
        }
    }
    if (live) {
        if (base > cap) { if (wrote0) list[0] = mark; }
        else for (int j = base + gl; j < cap; j += G) list[j] = kColsNone;
    }
    return base;   // #{p > 0} of the row (the mark and the support mask are decided on this count)
}


// z: in = raw row (padding = -inf), out = p.  Returns supp_size, tau (shifted frame).
// NW = waves holding the row (1: a lane group of one wave; then G < 64 skips the candidate path).
// cand: LDS, NW == 1: 2*kCandCap words private to this wave (values, columns); NW > 1: NW*64
// floats + NW ints + NW*64 ints + NW*64 floats (values, counts, columns, lane maxima).
// list != null: the row's support list is wanted; returns true if it was written
// here (from the candidates; n_listed = the #{p > 0} it saw), false if the caller has to scan the
// registers.
template <int E, int NW, bool CAND, bool VEC, class Red>
__device__ __forceinline__ bool sparsemax_row(float (&z)[E], Red& red, float* cand, int K, int& supp,
                                              float& tau_out, int& n_listed,
                                              int32_t* list = nullptr, int cap = 0,
                                              bool live = true, int mark = kColsOverflow,
                                              long long vdelta = 0)
{
    const float floor_k = cand_floor(K);
    const float margin = kLazyMargin;
    float mx = z[0];
#pragma unroll
    for (int e = 1; e < E; ++e) mx = fmaxf(mx, z[e]);
    float lane_mx = mx;
    mx = red.max(mx);
    lane_mx -= mx;   // this lane's largest element in the shifted frame (-inf for padding lanes)
#pragma unroll
    for (int e = 0; e < E; ++e) z[e] -= mx;

    int s_final = 1;
    double S_final = 0.0;
    bool fast = false;
    // what the lazy steps leave for a full-width search: the bound they reached, and whether they
    // stopped AT Michelot's fixed point (then the search starts with the edge, no iteration)
    float full_tau = -1.0f;
    bool full_fixed = false;
    bool solved = false;   // s_final / S_final already hold the answer (few-valued rows, NW > 1)
    if constexpr (CAND) {   // K <= 65536 < kCandMaxK in every kernel that comes through here
        const int lane = threadIdx.x & 63;
        float c[kCandPerLane];
        int cc[kCandPerLane];
        float thr_used = floor_k, tau_used = -1.0f;   // what the final candidate list was cut at
        if constexpr (NW == 1) {
            int* ccols = list ? reinterpret_cast<int*>(cand + kCandCap) : nullptr;
            const int colbase = VEC ? lane * 4 : lane, colstep = VEC ? 256 : 64;
            int n_c = compact_candidates<E, VEC>(z, floor_k, cand, kCandCap, ccols, colbase, colstep);
            if (n_c > kCandCap) {
                float tau = subset_bound(lane_mx);
                // a cut that did not move selects the same (too many) candidates: not tried again
                if (fmaxf(tau - margin, floor_k) > thr_used) {
                    wave_lds_sync();
                    thr_used = fmaxf(tau - margin, floor_k);
                    tau_used = tau;
                    n_c = compact_candidates<E, VEC>(z, thr_used, cand, kCandCap, ccols, colbase, colstep);
                }
                int cnt_prev = 0x7fffffff;
                for (int pass = 0; n_c > kCandCap && pass < kLazyPasses; ++pass) {
                    int cn = 0;
                    float sm = 0.0f;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const bool in = z[e] > tau;
                        cn += in ? 1 : 0;
                        sm += in ? z[e] : 0.0f;
                    }
                    red.cs(cn, sm);
                    if (cn >= cnt_prev) { full_fixed = true; break; }   // fixed point, wide support: full-width search
                    cnt_prev = cn;
                    tau = fmaxf(tau, (sm - 1.0f) / (float)cn);
                    if (cn <= kLazyTry && fmaxf(tau - margin, floor_k) > thr_used) {
                        wave_lds_sync();
                        thr_used = fmaxf(tau - margin, floor_k);
                        tau_used = tau;
                        n_c = compact_candidates<E, VEC>(z, thr_used, cand, kCandCap, ccols, colbase,
                                                         colstep);
                        if (n_c <= kCandCap) break;
                    }
                }
                full_tau = tau;   // a lower bound of the threshold (the full-width search starts there)
            }
            fast = n_c <= kCandCap;
            if (fast) {
                wave_lds_sync();
#pragma unroll
                for (int q = 0; q < kCandPerLane; ++q) {
                    const bool have = lane + 64 * q < n_c;
                    c[q] = have ? cand[lane + 64 * q] : -INFINITY;
                    cc[q] = (have && list) ? ccols[lane + 64 * q] : 0;
                }
                wave_lds_sync();
            }
        } else {
            const int wave = wave_index();
            int* cnt = reinterpret_cast<int*>(cand + NW * 64);
            int* ccols = list ? cnt + NW : nullptr;   // NW*64 candidate columns after the counts
            // candidates above the floor first; if they do not fit (narrow logit distribution),
            // block-wide Michelot steps raise the bound and the compaction is tried again above it
            float thr = floor_k, tau = -1.0f, thr_tried = -INFINITY;
            bool first = true;
            int cnt_prev = 0x7fffffff, passes = 0;
            // every lane's maximum, parked for the refinement of the bound after a first failure
            float* lmx = cand + 2 * NW * 64 + NW;
            lmx[threadIdx.x] = lane_mx;
            for (;;) {
                // a wave in which every lane holds an element above the cut has at least 64 of them:
                // its 64-entry list would have no room to spare, so it does not even try (narrow
                // logit distributions: saves the 64 ballots of a compaction that cannot succeed)
                // (nor does it try the same cut twice)
                const bool crowded = __ballot(lane_mx > thr) == ~0ull || !(thr > thr_tried);
                thr_tried = thr;
                // (a wave in which no lane's maximum is above the cut has nothing to compact)
                const int n_w = crowded ? 65 : (__ballot(lane_mx > thr) == 0ull) ? 0
                                        : compact_candidates<E, VEC>(z, thr, cand + wave * 64, 64,
                                                                     ccols ? ccols + wave * 64 : nullptr,
                                                                     VEC ? (int)threadIdx.x * 4 : (int)threadIdx.x,
                                                                     VEC ? 256 * NW : 64 * NW);
                if (lane == 0) cnt[wave] = n_w;
                __syncthreads();
                int total = 0, worst = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const int k = cnt[w];
                    total += k;
                    worst = k > worst ? k : worst;
                }
                fast = worst <= 64 && total <= kCandCap;
                if (fast && wave == 0) {   // (wave 0 solves for the workgroup, below)
#pragma unroll
                    for (int q = 0; q < kCandPerLane; ++q) {
                        const int idx = lane + 64 * q;
                        float v = -INFINITY;
                        int vc = 0, pre = 0;
#pragma unroll
                        for (int w = 0; w < NW; ++w) {
                            const int k = cnt[w];
                            if (idx >= pre && idx < pre + k) {
                                v = cand[w * 64 + (idx - pre)];
                                if (ccols) vc = ccols[w * 64 + (idx - pre)];
                            }
                            pre += k;
                        }
                        c[q] = v;
                        cc[q] = vc;
                    }
                }
                __syncthreads();  // the lists are rewritten by the next attempt / the next row
                if (fast) { thr_used = thr; tau_used = tau; break; }
                if (solved) break;   // (the list of a solved row did not fit: it goes without)
                if (first) {
                    // first failure: the threshold of the subset made of all 64 NW lane maxima is a
                    // lower bound of the row's.  Every wave runs that Michelot iteration on its own
                    // copy of the maxima (NW values per lane out of LDS, the same values in the same
                    // lanes and therefore the same bits in every wave): no pass over the registers
                    // and no workgroup barrier (the workgroup-wide version of this loop cost a
                    // barrier per step, 5-6 of them per row).
                    first = false;
                    WaveRed<64> wred;
                    // Rows made of ONE or TWO distinct values (all-equal rows; one dominant value
                    // over a plateau within 1 of it: SURVEY.md 8(d) S1's adversarial rows) have a
                    // closed form and need no search.  Noticed here for free: their lane maxima take
                    // at most two values.  Verified on the registers (row minimum, then the counts of
                    // 0, of the minimum and of the finite elements: two passes, two barriers) -- only
                    // for such rows.  (One pass with the lane maxima's second value as the minimum
                    // was measured: all-equal rows 3.98 -> 3.87 ms, but the register allocation of
                    // the common path moved and N(0,1) rows went 3.20 -> 3.31 ms; not kept.)
                    // (Kept narrow on purpose -- the lane maxima are re-read from LDS, the three
                    // counts travel as one double: this kernel sits at its 128-register limit.)
                    {
                        float mnv = INFINITY;
#pragma unroll
                        for (int w = 0; w < NW; ++w) {
                            const float t = lmx[w * 64 + lane];
                            mnv = (t > -INFINITY) ? fminf(mnv, t) : mnv;
                        }
                        mnv = wred.min(mnv);
                        int others = 0;
#pragma unroll
                        for (int w = 0; w < NW; ++w) {
                            const float t = lmx[w * 64 + lane];
                            others += (t > -INFINITY && t != 0.0f && t != mnv) ? 1 : 0;
                        }
                        others = wred.sum(others);
                        if (__builtin_amdgcn_readfirstlane(others) == 0) {   // same bits in every wave
                            float lmn = INFINITY;
#pragma unroll
                            for (int e = 0; e < E; ++e) {
                                float ze = z[e];
                                asm volatile("" : "+v"(ze));
                                lmn = (ze > -INFINITY) ? fminf(lmn, ze) : lmn;
                            }
                            const float rmin = red.min(lmn);
                            int c0 = 0, cm = 0, cf = 0;
#pragma unroll
                            for (int e = 0; e < E; ++e) {
                                float ze = z[e];
                                asm volatile("" : "+v"(ze));   // (see RegSrc: keeps the compare masks short-lived)
                                c0 += ze == 0.0f ? 1 : 0;
                                cm += ze == rmin ? 1 : 0;
                                cf += ze > -INFINITY ? 1 : 0;
                            }
                            // zeros + 2^18 (elements at rmin) + 2^36 (finite elements): K <= 65536
                            const double packed = (double)((long long)c0 + ((long long)cm << 18) + ((long long)cf << 36));
                            const long long pk = (long long)red.sum(packed);
                            const int n0 = (int)(pk & 0x3ffffll), nm = (int)((pk >> 18) & 0x3ffffll);
                            const int fin = (int)(pk >> 36);
                            if (rmin == 0.0f) {            // the row's minimum is its maximum: all equal,
                                s_final = fin;             // every rank passes, S = 0
                                S_final = 0.0;
                                solved = true;
                            } else if (n0 + nm == fin) {   // n0 zeros on top of a tie group of nm at rmin
                                const int k = classify_group(n0, 0.0, rmin, nm);
                                int cnt = k > 0 ? nm : 0;
                                if (k == 0) {
                                    if (tie_group_is_exact(n0, 0.0, rmin, nm)) cnt = exact_tie_count(n0, 0.0, rmin, nm);
                                    else cnt = red.sum(count_ranks_part(n0, 0.0, rmin, nm, red.idx(), red.stride()));
                                }
                                s_final = n0 + cnt;
                                S_final = (double)cnt * (double)rmin;
                                solved = true;
                                // the plateau stays out and the support is a handful of zeros: one
                                // more turn of this loop compacts them (the elements above rmin)
                                // into the candidate lists, only so that the row gets its support
                                // list and its backward does not have to read p
                                // (only if the plateau's p is exactly 0 -- the epilogue's own
                                // expression -- so that the zeros are ALL the non-zeros of p)
                                if (list && cnt == 0 && n0 <= 64 && !(rmin - (-1.0f / (float)n0) > 0.0f)) {
                                    thr = rmin;
                                    continue;
                                }
                            }
                            if (solved) break;
                        }
                    }
                    // (ONE wave runs it and hands the bound over through LDS: run by all NW waves
                    // -- four to a SIMD, ~1000 instructions each for a narrow row -- it was what a
                    // narrow K = 65536 row paid over a plain one: 3.93 -> see DESIGN 3.1)
                    float* bound_slot = lmx + NW * 64;
                    if (wave == 0) {
                        float v[NW];
#pragma unroll
                        for (int w = 0; w < NW; ++w) v[w] = lmx[w * 64 + lane];
                        tau = -1.0f;
                        for (int cp = 0x7fffffff;;) {
                            int cn = 0;
                            float sm = 0.0f;
#pragma unroll
                            for (int w = 0; w < NW; ++w) {
                                const bool in = v[w] > tau;
                                cn += in ? 1 : 0;
                                sm += in ? v[w] : 0.0f;
                            }
                            wred.cs(cn, sm);
                            if (cn >= cp) break;
                            cp = cn;
                            tau = fmaxf(tau, (sm - 1.0f) / (float)cn);
                        }
                        if (lane == 0) *bound_slot = tau;
                    }
                    __syncthreads();
                    tau = *bound_slot;
                    thr = fmaxf(tau - margin, floor_k);
                    continue;
                }
                bool give_up = false;
                for (;;) {   // block-uniform control flow: every value below comes out of red
                    int cn = 0;
                    float sm = 0.0f;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const bool in = z[e] > tau;
                        cn += in ? 1 : 0;
                        sm += in ? z[e] : 0.0f;
                    }
                    red.cs(cn, sm);
                    if (cn >= cnt_prev) { give_up = true; full_fixed = true; break; }
                    if (++passes > kLazyPasses) { give_up = true; break; }
                    cnt_prev = cn;
                    tau = fmaxf(tau, (sm - 1.0f) / (float)cn);
                    if (cn <= kLazyTry) break;
                }
                full_tau = tau;
                if (give_up) break;
                thr = fmaxf(tau - margin, floor_k);
            }
        }
        if constexpr (NW == 1) {
            if (fast && !solved) {
                WaveRed<64> wred;
                int m_robust = 0;
                solve_core(c, wred, kCandCap, s_final, S_final, &m_robust);
                // a list cut above the floor must be shown to hold everything the oracle could count
                if (thr_used > floor_k && !lazy_list_is_complete(thr_used, tau_used, m_robust, K)) fast = false;
            }
            if (fast && list) {
                const float tau_c = ((float)S_final - 1.0f) / (float)s_final;
                n_listed = emit_cols_from_candidates<kCandPerLane, 64>(c, cc, tau_c, lane, 0, list, cap, live, mark, vdelta);
            }
        } else if (fast) {   // (workgroup-uniform: decided on the counts in LDS)
            // Wave 0 searches the candidates for the whole workgroup and leaves the result in LDS;
            // the other waves wait at one barrier.  (Every wave used to hold the candidates and run
            // the same search on them -- no barrier, but four waves to a SIMD: the CU spent four
            // times the search on it while its loads were done and its stores not yet started.)
            float* res = cand + 3 * NW * 64 + NW + 1;   // s, S lo, S hi, still-fast flag
            if (wave_index() == 0) {
                bool ok = true;
                if (!solved) {
                    WaveRed<64> wred;
                    int m_robust = 0;
                    solve_core(c, wred, kCandCap, s_final, S_final, &m_robust);
                    if (thr_used > floor_k && !lazy_list_is_complete(thr_used, tau_used, m_robust, K)) ok = false;
                }
                if (ok && list) {
                    const float tau_c = ((float)S_final - 1.0f) / (float)s_final;
                    n_listed = emit_cols_from_candidates<kCandPerLane, 64>(c, cc, tau_c, lane, 0, list, cap, live, mark, vdelta);
                }
                if (lane == 0) {
                    res[0] = __int_as_float(s_final);
                    res[1] = __int_as_float(__double2loint(S_final));
                    res[2] = __int_as_float(__double2hiint(S_final));
                    res[3] = ok ? 1.0f : 0.0f;
                }
            }
            __syncthreads();
            s_final = __float_as_int(res[0]);
            S_final = __hiloint2double(__float_as_int(res[2]), __float_as_int(res[1]));
            fast = res[3] != 0.0f;
        }
    }
    if (__builtin_expect(!fast && !solved, 0)) solve_core(z, red, K, s_final, S_final, nullptr, full_tau, full_fixed);

    float c_s = (float)S_final - 1.0f;
    float tau_f = c_s / (float)s_final;
#pragma unroll
    for (int e = 0; e < E; ++e) z[e] = fmaxf(z[e] - tau_f, 0.0f);
    supp = s_final;
    tau_out = tau_f;
    return fast && list != nullptr;
}

// Same idea for rows held by a lane group of G < 64 lanes (K <= 512, several rows per wave): each
// group compacts its own candidates into its slice of the wave's LDS list (ballots are wave wide;
// a lane looks at its group's bits only) and the threshold search runs on Q values per lane
// instead of E.  If any row of the wave has more candidates than its slice holds, the whole wave
// takes the full-width search.
template <int E, int G, bool VEC>
__device__ __forceinline__ bool sparsemax_row_group(float (&z)[E], WaveRed<G>& red, float* cand, int K,
                                                    int& supp, float& tau_out, int& n_listed,
                                                    int32_t* list = nullptr, int cap = 0,
                                                    bool live = true, int mark = kColsOverflow,
                                                    long long vdelta = 0)
{
    const float floor_k = cand_floor(K);
    static_assert(G < 64, "lane groups inside one wave");
    constexpr int Q = 4;          // candidates per lane: capacity 4 G = 32, 64, 128 per row
    constexpr int CAP = G * Q;
    static_assert((64 / G) * CAP <= kCandCap, "the wave's LDS slice holds every group's list");
    float mx = z[0];
#pragma unroll
    for (int e = 1; e < E; ++e) mx = fmaxf(mx, z[e]);
    float lane_mx = mx;
    mx = red.max(mx);
    lane_mx -= mx;   // this lane's largest element in the shifted frame (-inf for padding lanes)
#pragma unroll
    for (int e = 0; e < E; ++e) z[e] -= mx;

    const int lane = threadIdx.x & 63, gl = lane % G, grp = lane / G;
    float* clist = cand + grp * CAP;
    int* ccols = reinterpret_cast<int*>(cand + kCandCap) + grp * CAP;
    float c[Q];
    int cc[Q];
    // The elements above thr as this row's candidates, if every row of the wave has at most CAP of
    // them (wave-uniform result).  Candidates per lane, then each lane's first slot by a scan over
    // the group's lanes: the per-element cost is a compare and an add (a wave-wide ballot per
    // element with the group's bits picked out of it cost three times as much, and these tiles are
    // instruction bound).  Nothing is written unless the lists fit.
    // Rows solved in closed form (few-valued rows, below) keep out of the lists: they show the
    // compaction their own cut thr_closed -- the plateau's value when the support is the handful of
    // zeros above it (the list is wanted for the support list), +inf otherwise -- and a row that
    // shows nothing gets one stand-in candidate (0 at column 0) so that the search on the lists has
    // a well-formed row to chew on; its result is replaced by the closed form.
    // (Their closed form waits in LDS, five words per lane behind the wave's lists: carried in
    // registers across the search it cost the common path 6 more spills and 8 % of its time.)
    bool closed = false;
    float* stash = cand + 2 * kCandCap + (threadIdx.x & 63);   // [k * 64]: thr, s, n, S lo, S hi
    auto compact_above = [&](float thr_all) -> bool {
        float thr = thr_all;
        if (__ballot(closed) != 0ull) thr = closed ? stash[0] : thr_all;
        int mine = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) mine += z[e] > thr ? 1 : 0;
        const int incl = group_scan_incl<G>(mine, gl);
        if (__ballot(incl > CAP) != 0ull) return false;   // incl of a group's last lane is its total
        const int total = __shfl(incl, lane | (G - 1));
        int off = incl - mine;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool in = z[e] > thr;
            if (__ballot(in) != 0ull) {  // wave-uniform
                if (in) {
                    clist[off] = z[e];
                    if (list) ccols[off] = col_of<E, VEC>(e, gl, G);
                }
                off += in ? 1 : 0;
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const bool have = gl + G * q < total;
            c[q] = have ? clist[gl + G * q] : -INFINITY;
            cc[q] = (have && list) ? ccols[gl + G * q] : 0;
        }
        if (closed && total == 0 && gl == 0) c[0] = 0.0f;
        wave_lds_sync();
        return true;
    };

    int s_final = 1;
    double S_final = 0.0;
    float tau = -1.0f, thr_used = floor_k, tau_used = -1.0f;
    bool all_fixed = false;   // wave-uniform: the Michelot iteration below has converged for every row
    bool fast = compact_above(floor_k);
    if (__builtin_expect(!fast, 0)) {
        // Rows made of ONE or TWO distinct values (all-equal rows; a few zeros over a plateau: the
        // adversarial rows of SURVEY.md 8(d) S1) have a closed form; left in, one such row sends
        // the other rows of its wave to the full-width search as well (a quarter of such rows among
        // N(0,1) rows at K = 256: 0.61 ms against 0.37-0.49 ms for any one kind alone).  Their lane
        // maxima take at most two values, which costs two group reductions to see; only then are
        // the registers counted (minimum; zeros, elements at the minimum, finite elements).
        const bool fin_l = lane_mx > -INFINITY;
        const float lmin = red.min(fin_l ? lane_mx : INFINITY);
        const int others = red.sum((fin_l && lane_mx != 0.0f && lane_mx != lmin) ? 1 : 0);
        if (__ballot(others == 0) != 0ull) {   // wave-uniform
            float lmn = INFINITY;
#pragma unroll
            for (int e = 0; e < E; ++e) lmn = (z[e] > -INFINITY) ? fminf(lmn, z[e]) : lmn;
            const float rmin = red.min(lmn);
            int pk = 0;   // zeros + 2^10 (elements at rmin) + 2^20 (finite elements): K <= 512
#pragma unroll
            for (int e = 0; e < E; ++e)
                pk += (z[e] == 0.0f ? 1 : 0) + (z[e] == rmin ? 1 << 10 : 0) + (z[e] > -INFINITY ? 1 << 20 : 0);
            pk = red.sum(pk);
            const int n0 = pk & 1023, nm = (pk >> 10) & 1023, fin = pk >> 20;
            const bool one = rmin == 0.0f;                    // the minimum is the maximum
            const bool two = !one && n0 + nm == fin;          // n0 zeros over a tie group of nm at rmin
            int cnt = 0;
            bool open = false;
            if (two) {
                const int k = classify_group(n0, 0.0, rmin, nm);
                cnt = k > 0 ? nm : 0;
                if (k == 0) {
                    if (tie_group_is_exact(n0, 0.0, rmin, nm)) cnt = exact_tie_count(n0, 0.0, rmin, nm);
                    else open = true;
                }
            }
            if (__ballot(open) != 0ull) {   // ranks inside the rounding band: counted by the row's lanes
                const int part = red.sum(open ? count_ranks_part(n0, 0.0, rmin, nm, gl, G) : 0);
                if (open) cnt = part;
            }
            int s_closed = 1, n_closed = 0;
            double S_closed = 0.0;
            float thr_closed = INFINITY;
            if (one) {
                s_closed = fin;   // every rank passes, S = 0, p = 1/K everywhere
                n_closed = fin;
            } else if (two) {
                s_closed = n0 + cnt;
                S_closed = (double)cnt * (double)rmin;
                // the plateau's p, by the epilogue's own expression
                const float pv = rmin - ((float)S_closed - 1.0f) / (float)s_closed;
                n_closed = n0 + (pv > 0.0f ? nm : 0);
                if (!(pv > 0.0f) && n0 <= CAP) thr_closed = rmin;   // the zeros are ALL the non-zeros of p
            }
            // a support list, if wanted, must come out right: the zeros as candidates, or a count
            // above cap (the row is marked, not listed); any other few-valued row goes the old way
            closed = (one || two) && (list == nullptr || thr_closed < INFINITY || n_closed > cap);
            if (__ballot(closed) != 0ull) {
                stash[0] = thr_closed;
                stash[64] = __int_as_float(s_closed);
                stash[128] = __int_as_float(n_closed);
                stash[192] = __int_as_float(__double2loint(S_closed));
                stash[256] = __int_as_float(__double2hiint(S_closed));
                fast = compact_above(floor_k);   // (own lane's words only: no synchronisation)
            }
        }
    }
    if (__builtin_expect(!fast, 0)) {
        // Narrow logit distributions.  Michelot starts from the threshold bound of the subset made
        // of one element per lane, its maximum -- (sum - 1) / count of them is below that subset's
        // threshold, which is below the row's -- and after every full-width step the candidates
        // are tried again above the current bound minus the lazy margin (see kLazyMargin: the
        // list is checked for completeness after the search on it).
        int cn = lane_mx > -INFINITY ? 1 : 0;
        float sm = lane_mx > -INFINITY ? lane_mx : 0.0f;
        red.cs(cn, sm);
        tau = fmaxf(-1.0f, (sm - 1.0f) / (float)cn);
        int cnt_prev = 0x7fffffff;
        for (int pass = 0; pass < kLazyPasses; ++pass) {
            thr_used = fmaxf(tau - kLazyMargin, floor_k);
            tau_used = tau;
            if (compact_above(thr_used)) { fast = true; break; }
            cn = 0;
            sm = 0.0f;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool in = z[e] > tau;
                cn += in ? 1 : 0;
                sm += in ? z[e] : 0.0f;
            }
            red.cs(cn, sm);
            const bool moving = cn < cnt_prev;
            // every row of the wave is at its fixed point, or one of them is there with more
            // elements above its threshold than a list holds (a wide support): full-width search
            all_fixed = __ballot(moving) == 0ull;
            if (all_fixed || __ballot(!closed && !moving && cn > CAP) != 0ull) break;
            if (moving) {
                cnt_prev = cn;
                tau = fmaxf(tau, (sm - 1.0f) / (float)cn);
            }
        }
    }
    bool listed = false;
    if (fast) {
        int m_robust = 0;
        solve_core(c, red, CAP, s_final, S_final, &m_robust, closed ? -1.0f : tau);
        // a list cut above the floor must be shown to hold everything the oracle could count
        const bool complete = closed || !(thr_used > floor_k) || lazy_list_is_complete(thr_used, tau_used, m_robust, K);
        if (__ballot(!complete) != 0ull) fast = false;
    }
    if (!fast) solve_core(z, red, K, s_final, S_final, nullptr, tau, all_fixed);
    const bool any_closed = __ballot(closed) != 0ull;
    if (any_closed && closed) {
        s_final = __float_as_int(stash[64]);
        S_final = __hiloint2double(__float_as_int(stash[256]), __float_as_int(stash[192]));
    }
    if (fast && list) {
        const float tau_c = ((float)S_final - 1.0f) / (float)s_final;
        n_listed = emit_cols_from_candidates<Q, G>(c, cc, tau_c, gl, grp, list, cap, live, mark, vdelta);
        if (any_closed && closed && !(stash[0] < INFINITY)) {   // nothing listed (but the stand-in): more than cap non-zeros
            n_listed = __float_as_int(stash[128]);
            if (live && gl == 0) list[0] = mark;
        }
        listed = true;
    }
    const float c_s = (float)S_final - 1.0f;
    const float tau_f = c_s / (float)s_final;
#pragma unroll
    for (int e = 0; e < E; ++e) z[e] = fmaxf(z[e] - tau_f, 0.0f);
    supp = s_final;
    tau_out = tau_f;
    return listed;
}

template <int E, class Red>
__device__ __forceinline__ void row_epilogue(const float (&p)[E], Red& red, bool want_nnz,
                                             bool want_ent, int& nnz, float& ent)
{
    if (want_nnz) {
        int c = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) c += (p[e] > 0.0f) ? 1 : 0;
        nnz = red.sum(c);
    }
    if (want_ent) {  // lvmhelpers/marg.py:18-28
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (p[e] > 0.0f) {
                float pc = fminf(fmaxf(p[e], FLT_EPSILON), 1.0f - FLT_EPSILON);
                acc += (double)(pc * logf(pc));
            }
        }
        ent = -(float)red.sum(acc);
    }
}

// Streaming (non-temporal) 16-byte accesses: every element is read once and written once, so
// the lines should not linger in L2 / Infinity Cache (+8 % on a plain tile copy on MI355X).
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p)
{
    f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float4* p, float4 v)
{
    f32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(p));
}

// ------------------------------------------------------------------------------- row <-> regs
// Lane `gl` of a group of GT lanes owns, for chunk i, the float4 with index i*GT + gl (VEC) or
// the element i*GT + gl (scalar): every load/store instruction covers consecutive addresses.
// FULL: K equals the tile capacity GT*E, so no bounds predicates are needed.
// UNI (rows that are not 16-byte addressable only): `row` is wave-uniform, so the scalar-layout
// accesses go through a raw buffer resource of K*4 bytes -- the hardware drops out-of-range lanes
// (loads return 0, stores vanish), which removes the per-element branches (each conditional load
// used to get its own s_waitcnt vmcnt(0)) and the per-element 64-bit addresses (the 1024-thread
// kernels spilled 140 registers on them): one 32-bit offset per access.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float* row, int K)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, K * 4, 0x00020000);
}

template <int E, bool VEC, bool FULL, bool UNI = false>
__device__ __forceinline__ void load_row(const float* __restrict__ row, int K, int gl, int GT,
                                         float (&z)[E], float pad)
{
    if constexpr (VEC) {
        const float4* r4 = reinterpret_cast<const float4*>(row);
        const int K4 = K >> 2;
#pragma unroll
        for (int i = 0; i < E / 4; ++i) {
            int f = i * GT + gl;
            float4 v = make_float4(pad, pad, pad, pad);
            // (uniform base of each group of 4 chunks) + (lane offset) + (immediate < 4 KB)
            if (FULL || f < K4)
                v = (UNI && !FULL) ? nt_load4((r4 + i * GT) + gl)   // ragged workgroup tile: scalar base per chunk
                        : nt_load4((r4 + (i & ~3) * GT) + ((i & 3) * GT + gl));
            z[4 * i + 0] = v.x; z[4 * i + 1] = v.y; z[4 * i + 2] = v.z; z[4 * i + 3] = v.w;
        }
    } else if constexpr (UNI) {
        const __amdgpu_buffer_rsrc_t rs = row_rsrc(row, K);
        // opaque copy of the lane index: keeps the 64 offsets and 64 lane masks derived from it
        // from being hoisted out of the kernel's row loop (and spilled); they cost one
        // instruction each where they are used
        asm volatile("" : "+v"(gl));
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int c = e * GT + gl;
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, c * 4, 0, 0));
            z[e] = (c < K) ? v : pad;
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            int c = e * GT + gl;
            z[e] = (c < K) ? row[c] : pad;
        }
    }
}

template <int E, bool VEC, bool FULL, bool UNI = false>
__device__ __forceinline__ void store_row(float* __restrict__ row, int K, int gl, int GT,
                                          const float (&z)[E])
{
    if constexpr (VEC) {
        float4* r4 = reinterpret_cast<float4*>(row);
        const int K4 = K >> 2;
#pragma unroll
        for (int i = 0; i < E / 4; ++i) {
            int f = i * GT + gl;
            if (FULL || f < K4)
                nt_store4((UNI && !FULL) ? (r4 + i * GT) + gl : (r4 + (i & ~3) * GT) + ((i & 3) * GT + gl),
                          make_float4(z[4 * i], z[4 * i + 1], z[4 * i + 2], z[4 * i + 3]));
        }
    } else if constexpr (UNI) {
        const __amdgpu_buffer_rsrc_t rs = row_rsrc(row, K);
        asm volatile("" : "+v"(gl));
#pragma unroll
        for (int e = 0; e < E; ++e)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, z[e]), rs, (e * GT + gl) * 4, 0, 0);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            int c = e * GT + gl;
            if (c < K) row[c] = z[e];
        }
    }
}

// The backward needs dy only where p != 0.  When the support is small the row of dy is not
// streamed at all: each non-zero of p fetches its own element (one 64/128-byte sector per support
// element instead of 4K bytes per row).  `nz` must be wave-uniformly known to be small by the
// caller; per register the ballot skips the load unless some lane has a non-zero there.
constexpr int kGatherMaxSupport = 96;   // ~ where 128-byte sectors per element reach a 16 KB row
// ... and for shorter rows where one 64-byte sector per element reaches the row itself
__device__ __forceinline__ int gather_max(int K)
{
    const int byrow = K >> 4;
    return byrow < kGatherMaxSupport ? byrow : kGatherMaxSupport;
}

// a[]: in = the row of p, out = the row of dx.  The gathered dy values replace the non-zeros of p
// in place (one register array; a 64-bit per-lane mask remembers where they are), zeros stay
// zeros, so the fp64 row sum adds the same numbers in the same order as the streaming path.
template <int E, bool VEC, class Red>
__device__ __forceinline__ void bwd_row_gather_bits(const float* __restrict__ dyrow, int gl, int GT,
                                                    float (&a)[E], uint32_t mlo, uint32_t mhi, int s,
                                                    Red& red);

template <int E, bool VEC, class Red>
__device__ __forceinline__ void bwd_row_gather(const float* __restrict__ dyrow, int gl, int GT,
                                               float (&a)[E], int s, Red& red)
{
    static_assert(E <= 64, "one mask bit per register");
    // mask bits with vector ALU only (no lane masks kept alive in scalar registers)
    uint32_t mlo = 0u, mhi = 0u;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        uint32_t t = __builtin_bit_cast(uint32_t, a[e]) & 0x7fffffffu;
        t = t < 1u ? t : 1u;
        if (e < 32) mlo |= t << e; else mhi |= t << (e - 32);
    }
    bwd_row_gather_bits<E, VEC>(dyrow, gl, GT, a, mlo, mhi, s, red);
}

// a[]: anything at the set bits (overwritten by the gathered dy), zero elsewhere on return
template <int E, bool VEC, class Red>
__device__ __forceinline__ void bwd_row_gather_bits(const float* __restrict__ dyrow, int gl, int GT,
                                                    float (&a)[E], uint32_t mlo, uint32_t mhi, int s,
                                                    Red& red)
{
    asm volatile("" : "+v"(mlo), "+v"(mhi));
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool nzero = ((e < 32 ? mlo >> e : mhi >> (e - 32)) & 1u) != 0u;
        if (__ballot(nzero) != 0ull) {   // wave-uniform
            const int col = VEC ? (((e >> 2) * GT + gl) * 4 + (e & 3)) : (e * GT + gl);
            if (nzero) a[e] = dyrow[col];
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) acc += (double)a[e];
    acc = red.sum(acc);
    const float v = (float)acc / (float)s;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool nzero = ((e < 32 ? mlo >> e : mhi >> (e - 32)) & 1u) != 0u;
        a[e] = nzero ? (a[e] - v) : 0.0f;
    }
}

// Register scan (rows that took the full-width threshold search): one pass; the lane that fills
// slot 0 overwrites it with the overflow mark if the row has more than cap non-zeros.
template <int E, int G, bool VEC>
__device__ __forceinline__ int emit_support_cols(const float (&p)[E], int gl, int grp,
                                                  int32_t* __restrict__ list, int cap, bool live,
                                                  int mark = kColsOverflow, long long vdelta = 0)
{
    if constexpr (G < 64) {
        // several rows per wave: count per lane, scan over the group's lanes, and only rows whose
        // support fits write columns (each lane its own, from its first slot on) -- a row that
        // overflows costs the count and the mark (these tiles are instruction bound)
        int mine = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) mine += p[e] > 0.0f ? 1 : 0;
        const int incl = group_scan_incl<G>(mine, gl);
        const int total = __shfl(incl, (int)(threadIdx.x & 63) | (G - 1));
        const bool fits = total <= cap;
        if (__ballot(fits && live) != 0ull) {
            int off = incl - mine;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool nz = p[e] > 0.0f;
                if (nz && fits && live) {
                    list[off] = col_of<E, VEC>(e, gl, G);
                    if (vdelta) *reinterpret_cast<float*>(reinterpret_cast<char*>(list + off) + vdelta) = p[e];
                }
                off += nz ? 1 : 0;
            }
        }
        if (live) {
            if (fits) for (int j = total + gl; j < cap; j += G) list[j] = kColsNone;
            else for (int j = gl; j < cap; j += G) list[j] = j == 0 ? mark : kColsNone;
        }
        return total;
    } else {
        int base = 0;
        bool wrote0 = false;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool nz = p[e] > 0.0f;
            const unsigned long long m = __ballot(nz);
            if (m != 0ull) {
                int cnt, below;
                group_count<G>(m, gl, grp, cnt, below);
                const int off = base + below;
                if (nz && live && off < cap) {
                    list[off] = col_of<E, VEC>(e, gl, G);
                    wrote0 = wrote0 || off == 0;
                    if (vdelta) *reinterpret_cast<float*>(reinterpret_cast<char*>(list + off) + vdelta) = p[e];
                }
                base += cnt;
            }
        }
        if (live) {
            if (base > cap) { if (wrote0) list[0] = mark; }
            else for (int j = base + gl; j < cap; j += G) list[j] = kColsNone;
        }
        return base;
    }
}

// dx of a row from its support list, in two rounds of stores: the whole row as zeros (no
// dependence on anything, issued before the list has even arrived), then -- once those stores
// have been acknowledged by the L2 (vmcnt == 0) -- the few values dy[col] - mean at the listed
// columns.  Both rounds come from the same wave, so they reach the same L2 in that order.  No
// register tile, no read of p, no streaming read of dy.  Slot j of the list is held by lane j % G
// of the row's lane group.
template <int E, int G, bool VEC, bool FULL>
__device__ __forceinline__ void bwd_row_from_cols(const float* __restrict__ dyrow,
                                                  float* __restrict__ dxrow,
                                                  const int32_t* __restrict__ list, int cap, int K,
                                                  int gl, int s, bool live, WaveRed<G>& red)
{
    constexpr int MAXS = kColsMaxCap / G;   // slots per lane
    int c[MAXS];
    float v[MAXS];
#pragma unroll
    for (int q = 0; q < MAXS; ++q) {
        const int j = gl + G * q;
        c[q] = (j < cap) ? list[j] : kColsNone;
    }
    if (live) {
        float zero[E];
#pragma unroll
        for (int e = 0; e < E; ++e) zero[e] = 0.0f;
        store_row<E, VEC, FULL>(dxrow, K, gl, G, zero);
    }
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < MAXS; ++q) {
        v[q] = 0.0f;
        if (c[q] >= 0) v[q] = dyrow[c[q]];
        acc += (double)v[q];
    }
    acc = red.sum(acc);
    const float mean = (float)acc / (float)s;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < MAXS; ++q)
        if (live && c[q] >= 0) dxrow[c[q]] = v[q] - mean;
}

// Small tiles (E <= 16): the row's dx is built in registers instead -- every entry of the list is
// broadcast through the lane group and placed with E compare-selects -- and stored once; partial
// writes into 1 KB rows cost more than they save.
template <int E, int G, bool VEC>
__device__ __forceinline__ void bwd_tile_from_cols(const float* __restrict__ dyrow,
                                                   const int32_t* __restrict__ list, int cap, int gl,
                                                   int lane, int s, WaveRed<G>& red, float (&a)[E])
{
    constexpr int MAXS = kColsMaxCap / G;
    int c[MAXS];
    float v[MAXS];
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < MAXS; ++q) {
        const int j = gl + G * q;
        c[q] = (j < cap) ? list[j] : kColsNone;
        v[q] = 0.0f;
        if (c[q] >= 0) v[q] = dyrow[c[q]];
        acc += (double)v[q];
    }
    acc = red.sum(acc);
    const float mean = (float)acc / (float)s;
#pragma unroll
    for (int e = 0; e < E; ++e) a[e] = 0.0f;
    const int gbase = lane - gl;
    const int col0 = col_of<E, VEC>(0, gl, G);
#pragma unroll
    for (int q = 0; q < MAXS; ++q) {
        for (int jj = 0; jj < G; ++jj) {
            const int col = __shfl(c[q], gbase + jj);
            if (__ballot(col >= 0) == 0ull) break;          // lists are packed: nothing further
            const float val = __shfl(v[q], gbase + jj) - mean;
            const int rel = col - col0;                      // negative for the sentinels
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int want = VEC ? ((e >> 2) * G * 4 + (e & 3)) : e * G;
                a[e] = (col >= 0 && rel == want) ? val : a[e];
            }
        }
    }
}

// dx of a row whose non-zero pattern the forward left as one bit per register (kColsMask), support
// small enough to gather.  Walking the pattern register by register costs one masked load
// instruction per occupied register slot -- ~38 per row at 58 non-zeros, one or two live lanes
// each, and the CU's one address unit takes them one after the other.  Here the pattern is first
// compacted into a column list in LDS (lane-exclusive prefix of the popcounts, every lane then
// writes its own few columns), the whole list is gathered with one or two full-lane loads, and dx
// is assembled 1024 columns at a time in an LDS tile (zero but for the scattered values) that is
// read back as float4s and stored -- every byte of dx written once, with full-width stores.
// wl: this wave's LDS (kPatLdsWords words).  Returns false (nothing written) if the pattern holds
// more than kPatList columns: the caller falls back to the register walk.
constexpr int kPatList = 128;                      // list slots (two per lane)
constexpr int kPatTile = 1024;                     // columns assembled per step
constexpr int kPatLdsWords = kPatTile + kPatList;  // per wave
template <int E>
__device__ __forceinline__ void bwd_row_assemble(const float* __restrict__ dyrow, float* __restrict__ dxrow,
                                                 int c0, int c1, int s, int K, int lane, bool live,
                                                 WaveRed<64>& red, float* wl, int col_base = 0);
template <int E>
__device__ __forceinline__ bool bwd_row_from_pattern(const float* __restrict__ dyrow, float* __restrict__ dxrow,
                                                     uint2 m, int s, int K, int lane, bool live,
                                                     WaveRed<64>& red, float* wl)
{
    static_assert(E % 16 == 0, "whole 1024-column steps");
    float* tile = wl;
    int* lcols = reinterpret_cast<int*>(wl + kPatTile);
    const int cnt = __popc(m.x) + __popc(m.y);
    int incl = group_scan_incl<16>(cnt, lane & 15);
    const int t0 = __builtin_amdgcn_readlane(incl, 15), t1 = __builtin_amdgcn_readlane(incl, 31),
              t2 = __builtin_amdgcn_readlane(incl, 47);
    incl += (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total > kPatList) return false;   // wave-uniform
    // zero tile (4 float4 per lane) and this lane's columns, in register order
    float4* tile4 = reinterpret_cast<float4*>(tile);
#pragma unroll
    for (int i = 0; i < kPatTile / 256; ++i) tile4[i * 64 + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    unsigned long long bits = ((unsigned long long)m.y << 32) | m.x;
    int off = incl - cnt;
    while (__ballot(bits != 0ull) != 0ull) {
        if (bits != 0ull) {
            const int e = __ffsll((long long)bits) - 1;
            bits &= bits - 1ull;
            lcols[off++] = (((e >> 2) * 64 + lane) << 2) + (e & 3);   // col_of<E, true>(e, lane, 64)
        }
    }
    wave_lds_sync();
    const int c0 = lane < total ? lcols[lane] : -1;
    const int c1 = lane + 64 < total ? lcols[lane + 64] : -1;
    bwd_row_assemble<E>(dyrow, dxrow, c0, c1, s, K, lane, live, red, wl);
    return true;
}

// (second half of the above, also usable with a list the forward wrote: c0 / c1 = this lane's
// columns or -1; the tile must be all zero on entry and is left all zero.  The E columns from
// col_base on are written -- one 4096-column segment of a wider row -- while the mean is taken
// over the whole list.)
template <int E>
__device__ __forceinline__ void bwd_row_assemble(const float* __restrict__ dyrow, float* __restrict__ dxrow,
                                                 int c0, int c1, int s, int K, int lane, bool live,
                                                 WaveRed<64>& red, float* wl, int col_base)
{
    float* tile = wl;
    float4* tile4 = reinterpret_cast<float4*>(tile);
    float v0 = 0.0f, v1 = 0.0f;
    if (c0 >= 0) v0 = dyrow[c0];
    if (c1 >= 0) v1 = dyrow[c1];
    const double acc = red.sum((double)v0 + (double)v1);
    const float mean = (float)acc / (float)s;
    v0 -= mean;
    v1 -= mean;
    float4* dx4 = reinterpret_cast<float4*>(dxrow);
    const int K4 = K >> 2;
#pragma unroll
    for (int q = 0; q < E / 16; ++q) {
        const int r0 = c0 - col_base, r1 = c1 - col_base;   // (columns below the base shift to negative)
        const bool h0 = c0 >= 0 && (r0 >> 10) == q, h1 = c1 >= 0 && (r1 >> 10) == q;
        if (h0) tile[r0 & (kPatTile - 1)] = v0;
        if (h1) tile[r1 & (kPatTile - 1)] = v1;
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < kPatTile / 256; ++i) {
            const float4 t = tile4[i * 64 + lane];
            const int f = (col_base >> 2) + q * (kPatTile / 4) + i * 64 + lane;   // float4 index in the row
            if (live && f < K4) nt_store4(dx4 + f, t);
        }
        wave_lds_sync();
        if (h0) tile[r0 & (kPatTile - 1)] = 0.0f;
        if (h1) tile[r1 & (kPatTile - 1)] = 0.0f;
    }
}

// The same assembly with a weight per listed column (entmax15: w = sqrt(p)): the row's dx is
// w (dy - q) at the listed columns, q = sum dy w / sum w, zero elsewhere.
template <int E>
__device__ __forceinline__ void bwd_row_assemble_weighted(const float* __restrict__ dyrow,
                                                          float* __restrict__ dxrow, int c0, float w0,
                                                          int K, int lane, bool live, WaveRed<64>& red,
                                                          float* wl)
{
    float* tile = wl;
    float4* tile4 = reinterpret_cast<float4*>(tile);
    float v0 = 0.0f;
    if (c0 >= 0) v0 = dyrow[c0];
    const double A = red.sum((double)(v0 * w0)), B = red.sum((double)w0);
    const float q = (float)(A / B);
    v0 = (v0 - q) * w0;
    float4* dx4 = reinterpret_cast<float4*>(dxrow);
    const int K4 = K >> 2;
#pragma unroll
    for (int qq = 0; qq < E / 16; ++qq) {
        const bool h0 = c0 >= 0 && (c0 >> 10) == qq;
        if (h0) tile[c0 & (kPatTile - 1)] = v0;
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < kPatTile / 256; ++i) {
            const float4 t = tile4[i * 64 + lane];
            const int f = qq * (kPatTile / 4) + i * 64 + lane;   // float4 index in the row
            if (live && f < K4) nt_store4(dx4 + f, t);
        }
        wave_lds_sync();
        if (h0) tile[c0 & (kPatTile - 1)] = 0.0f;
    }
}

// ------------------------------------------------------------------------------- kernels
constexpr int kWavesPerBlock = 4;

// Workgroup b is dispatched to XCD b % 8: hand each XCD one contiguous eighth of the rows, so the
// lines one XCD's L2 and memory channels see are neighbours (+3 % on a plain copy of this shape,
// +7 % on the K = 4096 forward; only for one-row-per-wave tiles, small rows lose a little).
template <bool REMAP>
__device__ __forceinline__ unsigned xcd_block()
{
    if constexpr (!REMAP) return blockIdx.x;
    const unsigned nb = gridDim.x, b = blockIdx.x, per = nb >> 3;
    if (b >= (per << 3)) return b;
    // ... and start every XCD at a different phase of its eighth: with all eight walking their
    // eighths in step, their streams sit a multiple of (rows/8) x 16 KB apart and meet in the same
    // memory channels (any skew >= 1 block measured -2 to -5 % on the forward; 17 is arbitrary)
    const unsigned xcd = b & 7u;
    return xcd * per + ((b >> 3) + xcd * 17u) % per;
}
// Workgroups that fill a CU keep the plain order (neighbouring rows in flight together), but a
// workgroup's XCD is blockIdx % 8 whatever the XCDs are doing: rows whose cost has a period that
// divides 8 (bench.py's adversarial mix repeats every 4 rows) would land all the expensive ones on
// the same two XCDs and the launch takes as long as those (K = 65536: 6.2 ms for a mix whose kinds
// average 4.2 ms).  Rotating the rows of every aligned group of eight by the group's index hands
// each XCD every residue in turn; the same eight rows are in flight at the same time as before.
__device__ __forceinline__ int64_t rot8(int64_t row, int64_t R)
{
    if ((row | 7) >= R) return row;
    return (row & ~(int64_t)7) | ((row + (row >> 3)) & 7);
}
// First row of this lane's group.  One row per wave (G == 64): the wave index goes through
// readfirstlane so that the row, its base addresses and `live` are scalar -- loads and stores then
// use the SGPR-base + 32-bit-lane-offset form instead of sixteen 64-bit VGPR addresses per tensor
// (which the register allocator used to spill).
template <int G>
__device__ __forceinline__ int64_t wave_row(int lane)
{
    if constexpr (G == 64) {
        const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        return (int64_t)xcd_block<true>() * kWavesPerBlock + wave;
    } else {
        return ((int64_t)xcd_block<false>() * kWavesPerBlock + (threadIdx.x >> 6)) * (64 / G) + lane / G;
    }
}
// waves per SIMD we ask the register allocator for (E data registers + ~40 of state)
#ifndef SMLVM_FWD_WPE16
#define SMLVM_FWD_WPE16 6   // tuning knob (tools/exp_build.sh): waves per SIMD asked for E <= 16 tiles
#endif
// (with the optional outputs the 64-register tile does not fit 128 VGPRs: 23 spilled registers
// were 6 KB of scratch per wave, 18 % on top of the launch's HBM traffic -- three waves instead)
constexpr int fwd_waves_per_eu(int E, bool extra = false) { return E <= 16 ? SMLVM_FWD_WPE16 : (E <= 32 ? 5 : (extra ? 3 : 4)); }
constexpr int bwd_waves_per_eu(int E) { return E <= 32 ? 8 : 3; }
constexpr int bwd_cols_waves_per_eu(int E) { return E <= 32 ? 8 : 3; }

// EXTRA: any of the optional outputs (nnz, tau, entropy) requested.
template <int G, int E, bool VEC, bool FULL, bool EXTRA>
__global__ __launch_bounds__(64 * kWavesPerBlock, fwd_waves_per_eu(E, EXTRA)) void sparsemax_fwd_wave_kernel(
    const float* __restrict__ x, float* __restrict__ p, int32_t* __restrict__ supp,
    int32_t* __restrict__ nnz, float* __restrict__ tau, float* __restrict__ ent,
    int32_t* __restrict__ cols, int cap, uint32_t* __restrict__ mask, int64_t R, int K,
    long long vdelta)
{
    const int lane = threadIdx.x & 63;
    const int gl = lane % G;
    int64_t row = wave_row<G>(lane);
    constexpr bool MASKS = (G == 64) && VEC && (E >= 32);
    constexpr bool MASKS16 = VEC && (E == 16);   // 32 < K <= 1024: 16 bits per lane, see below
    const int mark = ((MASKS || MASKS16) && mask != nullptr) ? kColsMask : kColsOverflow;
    const bool live = row < R;
    if (!live) row = R - 1;  // keep every lane busy (DPP needs whole groups); results discarded
    constexpr bool CAND = (G == 64);
    constexpr bool GCAND = (G < 64) && (E >= 8);   // per-group candidate lists
    // per wave: kCandCap candidate values followed by kCandCap candidate columns
    // (+ 5 x 64 words per wave for the lane groups' closed forms, see sparsemax_row_group)
    constexpr int kCandWords = 2 * kCandCap + (GCAND ? 5 * 64 : 0);
    __shared__ float cand[(CAND || GCAND) ? kWavesPerBlock * kCandWords : 1];
    float z[E];
    load_row<E, VEC, FULL>(x + row * K, K, gl, G, z, -INFINITY);
    WaveRed<G> red;
    int s, n = 0;
    float t, h = 0.0f;
    int32_t* list = cols ? cols + row * cap : nullptr;
    bool listed;
    int n_listed = 0;   // #{p > 0} as counted by whoever wrote (or gave up on) the row's list
    if constexpr (GCAND)
        listed = sparsemax_row_group<E, G, VEC>(z, red, cand + (threadIdx.x >> 6) * kCandWords, K, s, t,
                                                n_listed, list, cap, live, mark, vdelta);
    else
        listed = sparsemax_row<E, 1, CAND, VEC>(z, red, cand + (CAND ? (threadIdx.x >> 6) * kCandWords : 0),
                                                K, s, t, n_listed, list, cap, live, mark, vdelta);
    if constexpr (EXTRA) row_epilogue(z, red, nnz != nullptr, ent != nullptr, n, h);
    if (cols && !listed) n_listed = emit_support_cols<E, G, VEC>(z, gl, lane / G, list, cap, live, mark, vdelta);
    if constexpr (MASKS) {
        // The row was marked kColsMask exactly when that count exceeded cap (wave-uniform here; NOT
        // supp_size, which counts passing ranks and can differ from #{p > 0} with ties at the
        // threshold): leave the pattern as one bit per register
        if (cols && mask && n_listed > cap && live) {
            uint32_t mlo = 0u, mhi = 0u;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t b = z[e] > 0.0f ? 1u : 0u;
                if (e < 32) mlo |= b << e; else mhi |= b << (e - 32);
            }
            uint2* mrow = reinterpret_cast<uint2*>(mask + row * kMaskWordsPerRow);
            mrow[lane] = make_uint2(mlo, mhi);
        }
    }
    if constexpr (MASKS16) {
        // Tiles of 16 registers: EVERY row leaves its pattern (16 bits per lane, 2 G bytes per row:
        // 32 B at K = 256), so a wave of the backward that holds a row without a list can take all
        // of its rows from the patterns and never reads p; the rows that overflow their list are
        // marked kColsMask exactly as above.
        if (cols && mask && live) {
            uint32_t bits = 0u;
#pragma unroll
            for (int e = 0; e < E; ++e) bits |= (z[e] > 0.0f ? 1u : 0u) << e;
            reinterpret_cast<uint16_t*>(mask)[row * G + gl] = (uint16_t)bits;
        }
    }
    if (live) {
        store_row<E, VEC, FULL>(p + row * K, K, gl, G, z);
        if (gl == 0) {
            supp[row] = s;
            if constexpr (EXTRA) {
                if (nnz) nnz[row] = n;
                if (tau) tau[row] = t;
                if (ent) ent[row] = h;
            }
        }
    }
}

// Four waves per SIMD (128 VGPRs): with the 136-146 registers the allocator takes when left alone,
// a CU holds ONE workgroup of 2-8 waves and nothing overlaps its threshold search (K = 32768
// forward 1.82 -> 1.61 ms, narrow logits 2.66 -> 1.90 ms).
template <int NW, int E, bool VEC, bool FULL, bool EXTRA>
__global__ __launch_bounds__(64 * NW, 4) void sparsemax_fwd_block_kernel(
    const float* __restrict__ x, float* __restrict__ p, int32_t* __restrict__ supp,
    int32_t* __restrict__ nnz, float* __restrict__ tau, float* __restrict__ ent,
    int32_t* __restrict__ cols, int cap, int64_t R, int K)
{
    __shared__ unsigned long long lds[BlockRed<NW>::kLdsWords];
    __shared__ float cand[3 * NW * 64 + NW + 8];   // candidate values, counts, columns, lane maxima, subset bound, wave 0's result
    // XCD-aware row order for the small workgroups (K = 16384 forward 1.45 -> 1.37 ms; neutral at
    // NW = 8, a loss at NW = 16 where one workgroup fills the CU)
    for (int64_t row0 = xcd_block<(NW <= 4)>(); row0 < R; row0 += gridDim.x) {
        const int64_t row = NW <= 4 ? row0 : rot8(row0, R);
        float z[E];
        load_row<E, VEC, FULL, true>(x + row * K, K, threadIdx.x, 64 * NW, z, -INFINITY);
        BlockRed<NW> red(lds);
        int s, n = 0, n_listed = 0;
        float t, h = 0.0f;
        const bool listed = sparsemax_row<E, NW, true, VEC>(z, red, cand, K, s, t, n_listed,
                                                            cols ? cols + row * cap : nullptr, cap, true);
        if constexpr (EXTRA) row_epilogue(z, red, nnz != nullptr, ent != nullptr, n, h);
        store_row<E, VEC, FULL, true>(p + row * K, K, threadIdx.x, 64 * NW, z);
        if (threadIdx.x == 0) {
            supp[row] = s;
            if (cols && !listed) cols[row * cap] = kColsOverflow;   // full-width search: no list
            if constexpr (EXTRA) {
                if (nnz) nnz[row] = n;
                if (tau) tau[row] = t;
                if (ent) ent[row] = h;
            }
        }
        __syncthreads();  // LDS slots are reused by the next row
    }
}

template <int E, class Red>
__device__ __forceinline__ void bwd_row(float (&g)[E], const float (&pv)[E], int s, Red& red)
{
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) acc += (pv[e] != 0.0f) ? (double)g[e] : 0.0;
    acc = red.sum(acc);
    const float v = (float)acc / (float)s;
#pragma unroll
    for (int e = 0; e < E; ++e) g[e] = (pv[e] != 0.0f) ? (g[e] - v) : 0.0f;
}

template <int G, int E, bool VEC, bool FULL>
__global__ __launch_bounds__(64 * kWavesPerBlock, bwd_waves_per_eu(E)) void sparsemax_bwd_wave_kernel(
    const float* __restrict__ dy, const float* __restrict__ p, const int32_t* __restrict__ supp,
    float* __restrict__ dx, int64_t R, int K)
{
    const int lane = threadIdx.x & 63;
    const int gl = lane % G;
    int64_t row = wave_row<G>(lane);
    const bool live = row < R;
    if (!live) row = R - 1;
    const int s = supp[row];
    WaveRed<G> red;
    if (__ballot(s > gather_max(K)) == 0ull) {   // every row of this wave has a small support
        float a[E];
        load_row<E, VEC, FULL>(p + row * K, K, gl, G, a, 0.0f);
        bwd_row_gather<E, VEC>(dy + row * K, gl, G, a, s, red);
        if (live) store_row<E, VEC, FULL>(dx + row * K, K, gl, G, a);
    } else {
        float g[E], pv[E];
        load_row<E, VEC, FULL>(p + row * K, K, gl, G, pv, 0.0f);
        load_row<E, VEC, FULL>(dy + row * K, K, gl, G, g, 0.0f);
        bwd_row(g, pv, s, red);
        if (live) store_row<E, VEC, FULL>(dx + row * K, K, gl, G, g);
    }
}

// Backward from the forward's support lists: no read of p, no streaming read of dy; writes dx.
template <int G, int E, bool VEC, bool FULL>
__global__ __launch_bounds__(64 * kWavesPerBlock, bwd_cols_waves_per_eu(E)) void sparsemax_bwd_cols_wave_kernel(
    const float* __restrict__ dy, const int32_t* __restrict__ cols, int cap,
    const uint32_t* __restrict__ mask, const float* __restrict__ p, const int32_t* __restrict__ supp,
    float* __restrict__ dx, int64_t R, int K)
{
    const int lane = threadIdx.x & 63;
    const int gl = lane % G;
    int64_t row = wave_row<G>(lane);
    const bool live = row < R;
    if (!live) row = R - 1;
    const int s = supp[row];
    const int32_t* list = cols + row * cap;
    const int head = list[0];
    WaveRed<G> red;
    constexpr bool MASKS = (G == 64) && VEC && (E >= 32);
    constexpr bool SMALLTILE = VEC && (E <= 16) && (E >= 4);   // rows of <= 1024 columns, see below
    __shared__ __align__(16) float pat_lds[MASKS ? kWavesPerBlock * kPatLdsWords
                                                 : (SMALLTILE ? kWavesPerBlock * 64 * E : 1)];
    if constexpr (MASKS) {
        // no list, but the forward left the non-zero pattern (head and s are wave-uniform here)
        if (mask && head == kColsMask) {
            const uint2 m = reinterpret_cast<const uint2*>(mask + row * kMaskWordsPerRow)[lane];
            if (s <= gather_max(K) &&
                bwd_row_from_pattern<E>(dy + row * K, dx + row * K, m, s, K, lane, live, red,
                                        pat_lds + (threadIdx.x >> 6) * kPatLdsWords))
                return;
            float a[E];
            if (s <= gather_max(K)) {
#pragma unroll
                for (int e = 0; e < E; ++e) a[e] = 0.0f;
                bwd_row_gather_bits<E, VEC>(dy + row * K, gl, G, a, m.x, m.y, s, red);
            } else {
                // wider supports (a random 64-byte sector per element costs about as much as
                // 350 streamed bytes: measured break-even near 96 elements at K = 4096): stream dy
                // and select by the pattern -- same sum, same order as bwd_row, and still no p
                load_row<E, VEC, FULL>(dy + row * K, K, gl, G, a, 0.0f);
                double acc = 0.0;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool nz = ((e < 32 ? m.x >> e : m.y >> (e - 32)) & 1u) != 0u;
                    acc += nz ? (double)a[e] : 0.0;
                }
                acc = red.sum(acc);
                const float v = (float)acc / (float)s;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool nz = ((e < 32 ? m.x >> e : m.y >> (e - 32)) & 1u) != 0u;
                    a[e] = nz ? (a[e] - v) : 0.0f;
                }
            }
            if (live) store_row<E, VEC, FULL>(dx + row * K, K, gl, G, a);
            return;
        }
    }
    if constexpr (VEC && E == 16) {
        // some row of this wave has no list but the forward left every row's pattern: dy is
        // streamed (a row is at most 4 KB), p is not read; same sum in the same order as bwd_row
        if (mask && __ballot(head == kColsMask) != 0ull && __ballot(head == kColsOverflow) == 0ull) {
            const uint32_t bits = reinterpret_cast<const uint16_t*>(mask)[row * G + gl];
            float g[E];
            load_row<E, VEC, FULL>(dy + row * K, K, gl, G, g, 0.0f);
            double acc = 0.0;
#pragma unroll
            for (int e = 0; e < E; ++e) acc += ((bits >> e) & 1u) ? (double)g[e] : 0.0;
            acc = red.sum(acc);
            const float v = (float)acc / (float)s;
#pragma unroll
            for (int e = 0; e < E; ++e) g[e] = ((bits >> e) & 1u) ? (g[e] - v) : 0.0f;
            if (live) store_row<E, VEC, FULL>(dx + row * K, K, gl, G, g);
            return;
        }
    }
    if (__ballot(head < kColsNone) == 0ull) {   // every row of this wave has its list
        if constexpr (SMALLTILE) {
            // the wave's 64 / G rows side by side in its LDS tile (E G words each): zero, scatter
            // the listed dy - mean, read back as the float4s store_row writes
            constexpr int MAXS = kColsMaxCap / G;
            float* wl = pat_lds + (threadIdx.x >> 6) * (64 * E);
            float4* tile4 = reinterpret_cast<float4*>(wl);
#pragma unroll
            for (int i = 0; i < E / 4; ++i) tile4[i * 64 + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            int c[MAXS];
            float v[MAXS];
            double acc = 0.0;
            const float* dyrow = dy + row * K;
#pragma unroll
            for (int q = 0; q < MAXS; ++q) {
                const int j = gl + G * q;
                c[q] = (j < cap) ? list[j] : kColsNone;
                v[q] = 0.0f;
                if (c[q] >= 0) v[q] = dyrow[c[q]];
                acc += (double)v[q];
            }
            acc = red.sum(acc);
            const float mean = (float)acc / (float)s;
            float* myrow = wl + (lane / G) * (E * G);
#pragma unroll
            for (int q = 0; q < MAXS; ++q)
                if (c[q] >= 0) myrow[c[q]] = v[q] - mean;
            wave_lds_sync();
            float a[E];
            const float4* my4 = reinterpret_cast<const float4*>(myrow);
#pragma unroll
            for (int i = 0; i < E / 4; ++i) {
                const float4 t = my4[i * G + gl];
                a[4 * i] = t.x; a[4 * i + 1] = t.y; a[4 * i + 2] = t.z; a[4 * i + 3] = t.w;
            }
            if (live) store_row<E, VEC, FULL>(dx + row * K, K, gl, G, a);
        } else if constexpr (E <= 16) {
            float a[E];
            bwd_tile_from_cols<E, G, VEC>(dy + row * K, list, cap, gl, lane, s, red, a);
            if (live) store_row<E, VEC, FULL>(dx + row * K, K, gl, G, a);
        } else {
            if constexpr (MASKS) {
                // 16-byte rows of 1025 ... 4096 columns: dx is assembled in the wave's LDS tile and
                // written once with full-width stores (zero fill + scattered 4-byte stores and the
                // wait between them: 3.37 -> 2.9-3.0 ms at 1M x 4096, 4.13 -> 3.23 ms at 17 entries)
                float* wl = pat_lds + (threadIdx.x >> 6) * kPatLdsWords;
                float4* tile4 = reinterpret_cast<float4*>(wl);
#pragma unroll
                for (int i = 0; i < kPatTile / 256; ++i) tile4[i * 64 + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                const int c0 = lane < cap ? list[lane] : kColsNone;
                bwd_row_assemble<E>(dy + row * K, dx + row * K, c0, kColsNone, s, K, lane, live, red, wl);
            } else {
                bwd_row_from_cols<E, G, VEC, FULL>(dy + row * K, dx + row * K, list, cap, K, gl, s, live, red);
            }
        }
    } else if (__ballot(s > gather_max(K)) == 0ull) {
        float a[E];
        load_row<E, VEC, FULL>(p + row * K, K, gl, G, a, 0.0f);
        bwd_row_gather<E, VEC>(dy + row * K, gl, G, a, s, red);
        if (live) store_row<E, VEC, FULL>(dx + row * K, K, gl, G, a);
    } else {
        float g[E], pv[E];
        load_row<E, VEC, FULL>(p + row * K, K, gl, G, pv, 0.0f);
        load_row<E, VEC, FULL>(dy + row * K, K, gl, G, g, 0.0f);
        bwd_row(g, pv, s, red);
        if (live) store_row<E, VEC, FULL>(dx + row * K, K, gl, G, g);
    }
}

// Rows wider than one wave's tile (4096 < K <= 65536) with a support list: one wave per 4096-column
// segment of a row.  Every wave of a row reads the whole list and gathers the same <= 64 values
// (same fp64 sum in every wave), zero-fills its own segment and, once those stores are in the L2,
// stores the values whose columns fall into it -- no synchronisation between the waves of a row.
// Rows without a list are left to sparsemax_bwd_block_kernel (launched after this one).
constexpr int kSegTileMinSupport = 16;
template <bool VEC, bool FULL>
__global__ __launch_bounds__(64 * kWavesPerBlock) void sparsemax_bwd_cols_segment_kernel(
    const float* __restrict__ dy, const int32_t* __restrict__ cols, int cap,
    const int32_t* __restrict__ supp, float* __restrict__ dx, int64_t R, int K, int nseg)
{
    constexpr int SEG = 4096;
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware segment order: a write-only stream of 16 KB pieces runs at 6.4 TB/s with it and at
    // 5.2 TB/s without (tools/membench fill), this kernel 1.56 vs 1.70 ms at 32768 x 65536
    const int64_t wid = (int64_t)xcd_block<true>() * kWavesPerBlock + wave;
    if (wid >= R * nseg) return;
    const int64_t row = wid / nseg;
    const int seg = (int)(wid - row * nseg);
    const int c = lane < cap ? cols[row * cap + lane] : kColsNone;
    const int c2 = lane + 64 < cap ? cols[row * cap + lane + 64] : kColsNone;   // cap <= kColsMaxCapWide
    if (__builtin_amdgcn_readfirstlane(c) == kColsOverflow) return;
    float* dxrow = dx + row * K;
    const bool in1 = c >= 0 && (c >> 12) == seg, in2 = c2 >= 0 && (c2 >> 12) == seg;
    const bool mine = __ballot(in1 || in2) != 0ull;   // does a listed column fall into this segment?
    const int s_row = supp[row];
    if constexpr (VEC) {
        // Short lists keep the two rounds of stores below (the zero fill leaves at once, without
        // waiting for the gathers: 1.53 against 1.60 ms at 32768 x 65536, support 5); longer ones
        // are assembled in the wave's LDS tile and stored once (support 60-70: K = 65536 1.83 ->
        // 1.70 ms, K = 16384 2.28 -> 1.89 ms, K = 8196 0.79 -> 0.61 ms).
        if (mine && s_row > kSegTileMinSupport) {
            __shared__ __align__(16) float pat_lds[kWavesPerBlock * kPatLdsWords];
            float* wl = pat_lds + wave * kPatLdsWords;
            float4* tile4 = reinterpret_cast<float4*>(wl);
#pragma unroll
            for (int i = 0; i < kPatTile / 256; ++i) tile4[i * 64 + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            WaveRed<64> red;
            bwd_row_assemble<64>(dy + row * K, dxrow, c, c2, s_row, K, lane, true, red, wl, seg * SEG);
            return;
        }
    }
    {
        float zero[64];
#pragma unroll
        for (int e = 0; e < 64; ++e) zero[e] = 0.0f;
        if constexpr (VEC) {
            store_row<64, true, FULL>(dxrow + seg * SEG, K - seg * SEG, lane, 64, zero);
        } else {   // rows that are not 16-byte addressable: range-checked dword stores
            const __amdgpu_buffer_rsrc_t rs = row_rsrc(dxrow, K);
#pragma unroll
            for (int e = 0; e < 64; ++e)
                __builtin_amdgcn_raw_buffer_store_b32(0, rs, (seg * SEG + e * 64 + lane) * 4, 0, 0);
        }
    }
    if (!mine) return;   // an all-zero segment: nothing to gather, nothing to wait for
    const float v = c >= 0 ? dy[row * K + c] : 0.0f;
    const float v2 = c2 >= 0 ? dy[row * K + c2] : 0.0f;
    WaveRed<64> red;
    // (fp64 sum of <= 128 fp32 values by the wave's tree: differs from the streaming kernel's
    // column order by fp64 rounding only, far below the fp32 result)
    const double acc = red.sum((double)v + (double)v2);
    const float mean = (float)acc / (float)s_row;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (in1) dxrow[c] = v - mean;
    if (in2) dxrow[c2] = v2 - mean;
}

template <int NW, int E, bool VEC, bool FULL>
__global__ __launch_bounds__(64 * NW) void sparsemax_bwd_block_kernel(
    const float* __restrict__ dy, const float* __restrict__ p, const int32_t* __restrict__ supp,
    float* __restrict__ dx, int64_t R, int K, const int32_t* __restrict__ cols, int cap)
{
    __shared__ unsigned long long lds[BlockRed<NW>::kLdsWords];
    for (int64_t row0 = blockIdx.x; row0 < R; row0 += gridDim.x) {
        const int64_t row = rot8(row0, R);   // (see rot8: periodic row costs and the XCD round robin)
        if (cols && cols[row * cap] != kColsOverflow) continue;   // done from its list (block-uniform)
        const int s = supp[row];   // block-uniform
        BlockRed<NW> red(lds);
        if (s <= kGatherMaxSupport) {
            float a[E];
            load_row<E, VEC, FULL, true>(p + row * K, K, threadIdx.x, 64 * NW, a, 0.0f);
            bwd_row_gather<E, VEC>(dy + row * K, threadIdx.x, 64 * NW, a, s, red);
            store_row<E, VEC, FULL, true>(dx + row * K, K, threadIdx.x, 64 * NW, a);
        } else {
            float g[E], pv[E];
            load_row<E, VEC, FULL, true>(p + row * K, K, threadIdx.x, 64 * NW, pv, 0.0f);
            load_row<E, VEC, FULL, true>(dy + row * K, K, threadIdx.x, 64 * NW, g, 0.0f);
            bwd_row(g, pv, s, red);
            store_row<E, VEC, FULL, true>(dx + row * K, K, threadIdx.x, 64 * NW, g);
        }
        __syncthreads();
    }
}

// Entropy of an arbitrary [R,K] probability matrix (used when the normaliser is not ours):
// one wave per row, strided over K.
__global__ __launch_bounds__(256) void entropy_kernel(const float* __restrict__ p,
                                                      float* __restrict__ ent, int64_t R, int K)
{
    const int lane = threadIdx.x & 63;
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;  // whole wave leaves together
    const float* pr = p + row * K;
    double acc = 0.0;
    for (int c = lane; c < K; c += 64) {
        float v = pr[c];
        if (v > 0.0f) {
            float pc = fminf(fmaxf(v, FLT_EPSILON), 1.0f - FLT_EPSILON);
            acc += (double)(pc * logf(pc));
        }
    }
    acc = group_reduce<64>(acc, OpSum{});
    if (lane == 0) ent[row] = -(float)acc;
}

// ------------------------------------------------------------------------------- K > 65536
// Rows wider than one workgroup's registers (65536 elements) stay in memory: one 1024-thread block
// per row streams it -- max; the subset bound from the lane maxima; the candidates above it into
// per-wave LDS lists (the same <= 64 per wave, <= 256 per row as the block kernel), re-tried above
// a bound raised by streaming Michelot steps when they do not fit; the threshold search on the
// candidates in registers; p -- so a row is read three times (twice out of the L2 / Infinity Cache)
// and written once.  Rows whose support does not fit a candidate list run the whole search,
// edge passes included, as streaming passes.  Results are the same bits as the register kernels'.
constexpr int kStreamNW = 16;

template <bool VEC>
struct RowStream {
    const float* row;
    int K;
    float mx;
    // f(value in the shifted frame); every element once, in a fixed per-thread order
    template <class F>
    __device__ __forceinline__ void each(F&& f) const
    {
        if constexpr (VEC) {
            const float4* r4 = reinterpret_cast<const float4*>(row);
            const int K4 = K >> 2;
            for (int i = threadIdx.x; i < K4; i += 64 * kStreamNW) {
                const float4 v = r4[i];
                f(v.x - mx); f(v.y - mx); f(v.z - mx); f(v.w - mx);
            }
        } else {
            for (int c = threadIdx.x; c < K; c += 64 * kStreamNW) f(row[c] - mx);
        }
    }
};

// solve_core with the row behind a RowStream (same decisions, every pass re-reads the row)
template <class Src, class Red>
__device__ void solve_stream(const Src& src, Red& red, float tau, int nvis, int& s_out, double& S_out)
{
    int cnt_prev = 0x7fffffff;
    for (;;) {
        int c = 0;
        float sm = 0.0f;
        src.each([&](float v) { const bool in = v > tau; c += in ? 1 : 0; sm += in ? v : 0.0f; });
        red.cs(c, sm);
        if (c >= cnt_prev) break;
        cnt_prev = c;
        tau = fmaxf(tau, (sm - 1.0f) / (float)c);
    }
    solve_edge(src, red, tau, nvis, s_out, S_out);
}

template <bool VEC, bool EXTRA>
__global__ __launch_bounds__(64 * kStreamNW) void sparsemax_fwd_stream_kernel(
    const float* __restrict__ x, float* __restrict__ p, int32_t* __restrict__ supp,
    int32_t* __restrict__ nnz, float* __restrict__ tau_o, float* __restrict__ ent,
    int32_t* __restrict__ cols, int cap, int64_t R, int K)
{
    constexpr int NW = kStreamNW, NT = 64 * NW;
    __shared__ unsigned long long lds[BlockRed<NW>::kLdsWords];
    __shared__ float cand[2 * NW * 64 + NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int* cnt = reinterpret_cast<int*>(cand + NW * 64);
    int* ccols = cnt + NW;
    const int nstep = VEC ? (K >> 2) : K;   // float4 chunks or elements per row
    for (int64_t row0 = blockIdx.x; row0 < R; row0 += gridDim.x) {
        const int64_t row = rot8(row0, R);
        const float* xr = x + row * K;
        BlockRed<NW> red(lds);
        float lm = -INFINITY;
        {
            RowStream<VEC> raw{xr, K, 0.0f};
            raw.each([&](float v) { lm = fmaxf(lm, v); });
        }
        const float mx = red.max(lm);
        const float lane_mx = lm - mx;
        const RowStream<VEC> src{xr, K, mx};
        // lower bound of tau from the lane maxima: wave-local, then over all 1024 of them
        float tau = red.max(subset_bound(lane_mx));
        for (int cp = 0x7fffffff;;) {
            const bool in = lane_mx > tau;
            int cn = in ? 1 : 0;
            float sm = in ? lane_mx : 0.0f;
            red.cs(cn, sm);
            if (cn >= cp) break;
            cp = cn;
            tau = fmaxf(tau, (sm - 1.0f) / (float)cn);
        }
        const float floor_k = cand_floor(K < kCandMaxK ? K : kCandMaxK);
        const float margin = kLazyMargin;
        float thr = fmaxf(tau - margin, floor_k);
        float tau_used = tau;
        float c[kCandPerLane];
        int cc[kCandPerLane];
        bool fast = false, retried = false;
        int cnt_prev = 0x7fffffff, passes = 0;
        int s_final = 1;
        double S_final = 0.0;
        for (;;) {
            if (K > kCandMaxK) break;   // beyond the candidate floor's range: search the whole row
            int base = 0;
            for (int i0 = 0; i0 < nstep; i0 += NT) {   // block-uniform trip count
                const int i = i0 + threadIdx.x;
                const bool ok = i < nstep;
                float v[VEC ? 4 : 1];
                if constexpr (VEC) {
                    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) q = reinterpret_cast<const float4*>(xr)[i];
                    v[0] = q.x - mx; v[1] = q.y - mx; v[2] = q.z - mx; v[3] = q.w - mx;
                } else {
                    v[0] = (ok ? xr[i] : 0.0f) - mx;
                }
#pragma unroll
                for (int j = 0; j < (VEC ? 4 : 1); ++j) {
                    const bool in = ok && v[j] > thr;
                    const unsigned long long m = __ballot(in);
                    if (m != 0ull && base <= 64) {
                        const int off = base + lanes_below(m);
                        if (in && off < 64) {
                            cand[wave * 64 + off] = v[j];
                            ccols[wave * 64 + off] = VEC ? 4 * i + j : i;
                        }
                        base += __popcll(m);
                    }
                }
            }
            if (lane == 0) cnt[wave] = base;
            __syncthreads();
            int total = 0, worst = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int k = cnt[w];
                total += k;
                worst = k > worst ? k : worst;
            }
            fast = worst <= 64 && total <= kCandCap;
            if (fast) {
#pragma unroll
                for (int q = 0; q < kCandPerLane; ++q) {
                    const int idx = lane + 64 * q;
                    float vv = -INFINITY;
                    int vc = 0, pre = 0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        const int k = cnt[w];
                        if (idx >= pre && idx < pre + k) {
                            vv = cand[w * 64 + (idx - pre)];
                            vc = ccols[w * 64 + (idx - pre)];
                        }
                        pre += k;
                    }
                    c[q] = vv;
                    cc[q] = vc;
                }
            }
            __syncthreads();
            if (fast) {   // every wave solves the same list: uniform decisions, no barriers
                WaveRed<64> wred;
                int m_robust = 0;
                solve_core(c, wred, kCandCap, s_final, S_final, &m_robust);
                // a list cut above the floor must be shown to hold everything the oracle could count
                if (thr <= floor_k || lazy_list_is_complete(thr, tau_used, m_robust, K)) break;
                fast = false;
                if (retried) break;
                retried = true;   // once more, cut where the check asks (a short support under a wide row)
                thr = fmaxf(lazy_complete_thr(tau_used, m_robust, K), floor_k);
                continue;
            }
            bool give_up = false;
            for (;;) {
                int cn = 0;
                float sm = 0.0f;
                src.each([&](float v) { const bool in = v > tau; cn += in ? 1 : 0; sm += in ? v : 0.0f; });
                red.cs(cn, sm);
                if (cn >= cnt_prev || ++passes > kLazyPasses) { give_up = true; break; }
                cnt_prev = cn;
                tau = fmaxf(tau, (sm - 1.0f) / (float)cn);
                if (cn <= kLazyTry) break;
            }
            if (give_up) break;
            thr = fmaxf(tau - margin, floor_k);
            tau_used = tau;
        }
        int32_t* list = cols ? cols + row * cap : nullptr;
        if (fast) {
            if (list) {
                const float tau_c = ((float)S_final - 1.0f) / (float)s_final;
                emit_cols_from_candidates<kCandPerLane, 64>(c, cc, tau_c, lane, 0, list, cap,
                                                            threadIdx.x < 64);
            }
        } else {
            solve_stream(src, red, tau, K, s_final, S_final);
            if (list && threadIdx.x == 0) list[0] = kColsOverflow;
        }
        const float tau_f = ((float)S_final - 1.0f) / (float)s_final;
        int n = 0;
        double h = 0.0;
        auto out1 = [&](float xv) {
            const float pv = fmaxf((xv - mx) - tau_f, 0.0f);
            if constexpr (EXTRA) {
                n += pv > 0.0f ? 1 : 0;
                if (ent && pv > 0.0f) {
                    const float pc = fminf(fmaxf(pv, FLT_EPSILON), 1.0f - FLT_EPSILON);
                    h += (double)(pc * logf(pc));
                }
            }
            return pv;
        };
        float* pr = p + row * K;
        if constexpr (VEC) {
            const float4* r4 = reinterpret_cast<const float4*>(xr);
            float4* o4 = reinterpret_cast<float4*>(pr);
            for (int i = threadIdx.x; i < nstep; i += NT) {
                const float4 q = nt_load4(r4 + i);
                nt_store4(o4 + i, make_float4(out1(q.x), out1(q.y), out1(q.z), out1(q.w)));
            }
        } else {
            for (int i = threadIdx.x; i < K; i += NT) pr[i] = out1(xr[i]);
        }
        if constexpr (EXTRA) {
            if (nnz) n = red.sum(n);
            if (ent) h = red.sum(h);
        }
        if (threadIdx.x == 0) {
            supp[row] = s_final;
            if constexpr (EXTRA) {
                if (nnz) nnz[row] = n;
                if (tau_o) tau_o[row] = tau_f;
                if (ent) ent[row] = -(float)h;
            }
        }
        __syncthreads();
    }
}

// p-based backward of such rows (rows that have a support list are written by the segment kernel)
template <bool VEC>
__global__ __launch_bounds__(64 * kStreamNW) void sparsemax_bwd_stream_kernel(
    const float* __restrict__ dy, const float* __restrict__ p, const int32_t* __restrict__ supp,
    float* __restrict__ dx, int64_t R, int K, const int32_t* __restrict__ cols, int cap)
{
    constexpr int NW = kStreamNW, NT = 64 * NW;
    __shared__ unsigned long long lds[BlockRed<NW>::kLdsWords];
    for (int64_t row0 = blockIdx.x; row0 < R; row0 += gridDim.x) {
        const int64_t row = rot8(row0, R);
        if (cols && cols[row * cap] >= kColsNone) continue;   // block-uniform
        BlockRed<NW> red(lds);
        const float* pr = p + row * K;
        const float* gr = dy + row * K;
        float* o = dx + row * K;
        double acc = 0.0;
        if constexpr (VEC) {
            const float4* p4 = reinterpret_cast<const float4*>(pr);
            const float4* g4 = reinterpret_cast<const float4*>(gr);
            for (int i = threadIdx.x; i < (K >> 2); i += NT) {
                const float4 a = p4[i], g = g4[i];
                acc += a.x != 0.0f ? (double)g.x : 0.0;
                acc += a.y != 0.0f ? (double)g.y : 0.0;
                acc += a.z != 0.0f ? (double)g.z : 0.0;
                acc += a.w != 0.0f ? (double)g.w : 0.0;
            }
        } else {
            for (int i = threadIdx.x; i < K; i += NT) acc += pr[i] != 0.0f ? (double)gr[i] : 0.0;
        }
        acc = red.sum(acc);
        const float v = (float)acc / (float)supp[row];
        if constexpr (VEC) {
            const float4* p4 = reinterpret_cast<const float4*>(pr);
            const float4* g4 = reinterpret_cast<const float4*>(gr);
            float4* o4 = reinterpret_cast<float4*>(o);
            for (int i = threadIdx.x; i < (K >> 2); i += NT) {
                const float4 a = p4[i], g = g4[i];
                nt_store4(o4 + i, make_float4(a.x != 0.0f ? g.x - v : 0.0f, a.y != 0.0f ? g.y - v : 0.0f,
                                              a.z != 0.0f ? g.z - v : 0.0f, a.w != 0.0f ? g.w - v : 0.0f));
            }
        } else {
            for (int i = threadIdx.x; i < K; i += NT) o[i] = pr[i] != 0.0f ? gr[i] - v : 0.0f;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------- dispatch
struct FwdArgs {
    const float* x; float* p; int32_t* supp; int32_t* nnz; float* tau; float* ent;
    int64_t R; int K; hipStream_t st;
    bool vec, extra;
    int32_t* cols; int cap;
    uint32_t* mask;
    float* vals = nullptr;   // value lists (K <= 4096 only), slot for slot with cols
};
struct BwdArgs {
    const float* dy; const float* p; const int32_t* supp; float* dx;
    int64_t R; int K; hipStream_t st;
    bool vec;
    const int32_t* cols; int cap;
    const uint32_t* mask;
};

static unsigned wave_grid(int64_t R, int G)
{
    const int64_t rows_per_block = (int64_t)kWavesPerBlock * (64 / G);
    return (unsigned)((R + rows_per_block - 1) / rows_per_block);
}
static unsigned block_grid(int64_t R) { return (unsigned)(R < 65536 ? R : 65536); }


template <int G, int E, bool VEC, bool FULL, bool EXTRA>
static void fwd_wave_go(const FwdArgs& a)
{
    hipLaunchKernelGGL((sparsemax_fwd_wave_kernel<G, E, VEC, FULL, EXTRA>), dim3(wave_grid(a.R, G)),
                       dim3(64 * kWavesPerBlock), 0, a.st, a.x, a.p, a.supp, a.nnz, a.tau, a.ent,
                       a.cols, a.cap, a.mask, a.R, a.K,
                       (long long)((a.cols && a.vals) ? reinterpret_cast<const char*>(a.vals) -
                                                            reinterpret_cast<const char*>(a.cols) : 0));
}
template <int NW, int E, bool VEC, bool FULL, bool EXTRA>
static void fwd_block_go(const FwdArgs& a)
{
    hipLaunchKernelGGL((sparsemax_fwd_block_kernel<NW, E, VEC, FULL, EXTRA>), dim3(block_grid(a.R)),
                       dim3(64 * NW), 0, a.st, a.x, a.p, a.supp, a.nnz, a.tau, a.ent, a.cols, a.cap,
                       a.R, a.K);
}
template <int G, int E, bool VEC, bool FULL>
static void bwd_wave_go(const BwdArgs& a)
{
    if (a.cols)
        hipLaunchKernelGGL((sparsemax_bwd_cols_wave_kernel<G, E, VEC, FULL>), dim3(wave_grid(a.R, G)),
                           dim3(64 * kWavesPerBlock), 0, a.st, a.dy, a.cols, a.cap, a.mask, a.p, a.supp,
                           a.dx, a.R, a.K);
    else
        hipLaunchKernelGGL((sparsemax_bwd_wave_kernel<G, E, VEC, FULL>), dim3(wave_grid(a.R, G)),
                           dim3(64 * kWavesPerBlock), 0, a.st, a.dy, a.p, a.supp, a.dx, a.R, a.K);
}
static void launch_cols_segments(const BwdArgs& a);
template <int NW, int E, bool VEC, bool FULL>
static void bwd_block_go(const BwdArgs& a)
{
    if (a.cols) launch_cols_segments(a);   // listed rows first, one wave per 4096-column segment
    hipLaunchKernelGGL((sparsemax_bwd_block_kernel<NW, E, VEC, FULL>), dim3(block_grid(a.R)),
                       dim3(64 * NW), 0, a.st, a.dy, a.p, a.supp, a.dx, a.R, a.K, a.cols, a.cap);
}

// mode selection for one tile shape (A = lanes per row, or waves per row for the block kernels)
template <int A, int E>
static void launch_fwd_wave(const FwdArgs& a)
{
    const bool full = a.vec && a.K == A * E;
    if (!a.vec)      a.extra ? fwd_wave_go<A, E, false, false, true>(a) : fwd_wave_go<A, E, false, false, false>(a);
    else if (full)   a.extra ? fwd_wave_go<A, E, true, true, true>(a)   : fwd_wave_go<A, E, true, true, false>(a);
    else             a.extra ? fwd_wave_go<A, E, true, false, true>(a)  : fwd_wave_go<A, E, true, false, false>(a);
}
template <int A, int E>
static void launch_fwd_block(const FwdArgs& a)
{
    const bool full = a.vec && a.K == 64 * A * E;
    if (!a.vec)      a.extra ? fwd_block_go<A, E, false, false, true>(a) : fwd_block_go<A, E, false, false, false>(a);
    else if (full)   a.extra ? fwd_block_go<A, E, true, true, true>(a)   : fwd_block_go<A, E, true, true, false>(a);
    else             a.extra ? fwd_block_go<A, E, true, false, true>(a)  : fwd_block_go<A, E, true, false, false>(a);
}
template <int A, int E>
static void launch_bwd_wave(const BwdArgs& a)
{
    const bool full = a.vec && a.K == A * E;
    if (!a.vec)    bwd_wave_go<A, E, false, false>(a);
    else if (full) bwd_wave_go<A, E, true, true>(a);
    else           bwd_wave_go<A, E, true, false>(a);
}
template <int A, int E>
static void launch_bwd_block(const BwdArgs& a)
{
    const bool full = a.vec && a.K == 64 * A * E;
    if (!a.vec)    bwd_block_go<A, E, false, false>(a);
    else if (full) bwd_block_go<A, E, true, true>(a);
    else           bwd_block_go<A, E, true, false>(a);
}

static void launch_fwd_stream(const FwdArgs& a)
{
    const dim3 grid(block_grid(a.R)), block(64 * kStreamNW);
#define SMLVM_FWD_STREAM(V, X)                                                                  \
    hipLaunchKernelGGL((sparsemax_fwd_stream_kernel<V, X>), grid, block, 0, a.st, a.x, a.p, a.supp, \
                       a.nnz, a.tau, a.ent, a.cols, a.cap, a.R, a.K)
    if (a.vec) { if (a.extra) SMLVM_FWD_STREAM(true, true); else SMLVM_FWD_STREAM(true, false); }
    else       { if (a.extra) SMLVM_FWD_STREAM(false, true); else SMLVM_FWD_STREAM(false, false); }
#undef SMLVM_FWD_STREAM
}

static void launch_cols_segments(const BwdArgs& a)
{
    const int nseg = (a.K + 4095) / 4096;
    const int64_t waves = a.R * nseg;
    const dim3 grid((unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock));
    if (!a.vec)
        hipLaunchKernelGGL((sparsemax_bwd_cols_segment_kernel<false, false>), grid, dim3(64 * kWavesPerBlock),
                           0, a.st, a.dy, a.cols, a.cap, a.supp, a.dx, a.R, a.K, nseg);
    else if (a.K % 4096 == 0)
        hipLaunchKernelGGL((sparsemax_bwd_cols_segment_kernel<true, true>), grid, dim3(64 * kWavesPerBlock),
                           0, a.st, a.dy, a.cols, a.cap, a.supp, a.dx, a.R, a.K, nseg);
    else
        hipLaunchKernelGGL((sparsemax_bwd_cols_segment_kernel<true, false>), grid, dim3(64 * kWavesPerBlock),
                           0, a.st, a.dy, a.cols, a.cap, a.supp, a.dx, a.R, a.K, nseg);
}

static void launch_bwd_stream(const BwdArgs& a)
{
    if (a.cols) launch_cols_segments(a);
    const dim3 grid(block_grid(a.R)), block(64 * kStreamNW);
    if (a.vec)
        hipLaunchKernelGGL((sparsemax_bwd_stream_kernel<true>), grid, block, 0, a.st, a.dy, a.p, a.supp,
                           a.dx, a.R, a.K, a.cols, a.cap);
    else
        hipLaunchKernelGGL((sparsemax_bwd_stream_kernel<false>), grid, block, 0, a.st, a.dy, a.p, a.supp,
                           a.dx, a.R, a.K, a.cols, a.cap);
}

// Shape table: smallest register tile whose capacity covers K.
#define SMLVM_DISPATCH(WAVE, BLOCK, args)                                                       \
    do {                                                                                        \
        const int K_ = (args).K;                                                                \
        if (K_ <= 32) WAVE<8, 4>(args);                                                         \
        else if (K_ <= 128) WAVE<8, 16>(args);                                                  \
        else if (K_ <= 256) WAVE<16, 16>(args);                                                 \
        else if (K_ <= 512) WAVE<32, 16>(args);                                                 \
        else if (K_ <= 1024) WAVE<64, 16>(args);                                                \
        else if (K_ <= 2048) WAVE<64, 32>(args);                                                \
        else if (K_ <= 4096) WAVE<64, 64>(args);                                                \
        else if (K_ <= 8192) BLOCK<2, 64>(args);                                                \
        else if (K_ <= 16384) BLOCK<4, 64>(args);                                               \
        else if (K_ <= 32768) BLOCK<8, 64>(args);                                               \
        else BLOCK<16, 64>(args);                                                               \
    } while (0)

// ------------------------------------------------------------------------------- entmax15
// 1.5-entmax (the default normaliser of ExplicitWrapper, lvmhelpers/marg.py:46; third-party
// entmax.entmax15): p = max(z - tau, 0)^2 with z = x/2 - max(x/2) and tau the root of
// f(tau) = sum max(z - tau, 0)^2 - 1.  The package sorts the row; here the row stays in the same
// register tiles as sparsemax and tau comes from Newton's iteration started at -1: f is convex and
// decreasing and f(-1) >= 0 (z_max = 0), so the iterates rise monotonically to the root and never
// pass it -- every set {z > tau_k} contains the support.  When the step has shrunk below fp32
// resolution the root is recomputed in closed form (fp64) on the set that is left,
// tau = mean - sqrt((1 - ss)/c), the package's own formula for that prefix.  Floating-point
// parity (2e-6 on p), not bit parity: the package's fp32 cumsums are not reproduced.
constexpr int kE15MaxNewton = 64;

// One Newton step of f(tau) = sum max(v - tau, 0)^2 - 1 on the values v[] (fp32 partial sums per
// lane, fp64 across lanes); returns the step, and in `lower` a LOWER bound of the size of
// {v > tau}: (sum d)^2 <= count * sum d^2 (counting the set itself costs half a step more per
// element, and rows that never get a list -- narrow logit distributions -- would pay it every step).
template <int N, class Red>
__device__ __forceinline__ float entmax15_newton_step(const float (&v)[N], float tau, Red& red, float& lower)
{
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const float d = fmaxf(v[e] - tau, 0.0f);
        s1 += d;
        s2 = fmaf(d, d, s2);
    }
    double a = (double)s1, b = (double)s2;
    red.sum2(a, b);
    // b - 1 cancels and is taken in fp64; the quotients need no more than v_rcp_f32 gives (the
    // partial sums carry fp32 rounding already; an fp64 division per step was a fifth of the
    // lane-group kernels' iteration)
    const float af = (float)a;
    lower = af * af * __builtin_amdgcn_rcpf((float)b);
    return (float)(b - 1.0) * __builtin_amdgcn_rcpf(2.0f * af);
}

// the package's closed form on the set {v > tau}: tau = mean - sqrt((1 - ss)/c), fp64
template <int N, class Red>
__device__ __forceinline__ float entmax15_closed_form(const float (&v)[N], float tau, Red& red)
{
    int c = 0;
    double S1 = 0.0, S2 = 0.0;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const bool in = v[e] > tau;
        const double w = in ? (double)v[e] : 0.0;
        c += in ? 1 : 0;
        S1 += w;
        S2 += w * w;
    }
    red.sums(S1, S2, c);
    if (c > 0) {
        const double mean = S1 / (double)c, ss = S2 - (double)c * mean * mean;
        const double delta = (1.0 - ss) / (double)c;
        const float tc = (float)(mean - sqrt(delta > 0.0 ? delta : 0.0));
        tau = tc > tau ? tc : tau;                   // (never below the Newton iterate)
    }
    return tau;
}

// NW = waves that hold the row (0: a lane group of one wave; GC: with candidate lists per group).
// cand: LDS, NW == 1: kCandCap floats private to the wave; NW > 1: NW*64 floats + NW ints;
// NW == 0 with GC: kCandCap floats private to the wave, a slice of 4 G per lane group.
//
// Candidates (round 3).  The Newton iterates rise monotonically and never pass the root, so every
// set {z > tau_k} holds the support and the iteration from tau_k on only ever looks at that set.
// Once it has at most kCandCap elements (after one or two full-width steps for N(0,1) logits) it
// is compacted through LDS -- the same lists as sparsemax's threshold search -- and the remaining
// ~10 steps and the closed form run on 4 values per lane instead of 64, without workgroup barriers
// (every wave of a workgroup holds the same list).  Rows whose sets stay wide (narrow logit
// distributions) iterate full width as before.
// list != nullptr (NW == 1 only): the row's support list -- columns with p > 0, and at byte distance
// vdelta the values sqrt(p) = z - tau slot for slot (what the backward weighs with) -- comes from the
// candidates; rows that finished full width are marked kColsOverflow (their backward reads p).
template <int E, int NW, bool VEC, bool GC = false, class Red>
__device__ __forceinline__ void entmax15_row(float (&z)[E], Red& red, int& nnz_out, float& tau_out,
                                             float* cand = nullptr, int32_t* list = nullptr, int cap = 0,
                                             bool live = true, long long vdelta = 0)
{
    float mx = z[0];
#pragma unroll
    for (int e = 1; e < E; ++e) mx = fmaxf(mx, z[e]);
    mx = red.max(mx);
#pragma unroll
    for (int e = 0; e < E; ++e) z[e] = (z[e] - mx) * 0.5f;   // padding stays -inf
    float tau = -1.0f;
    bool listed = false;
    float c[kCandPerLane];
    int cc[kCandPerLane];
#pragma unroll
    for (int q = 0; q < kCandPerLane; ++q) cc[q] = 0;
    // attempt the compaction once the bound says the set may fit (for a tail of N(0,1) logits the
    // set is about twice the bound); after a failure, when the bound has halved
    float try_below = 0.4f * (float)kCandCap;
    int it = 0;
    for (; it < kE15MaxNewton; ++it) {
        float cnt;
        const float step = entmax15_newton_step(z, tau, red, cnt);
        const bool moving = step > 1e-7f;            // false for NaN rows too
        if (moving) tau += step;
        if (__ballot(moving) == 0ull) break;         // every row this wave holds has converged
        if constexpr (NW == 0 && GC) {
            // Several rows per wave: every lane group compacts its row's set into its slice of the
            // wave's list once EVERY row's bound says it may fit (the decision is the wave's: the
            // rows run one instruction stream), and the remaining steps look at 4 values per lane.
            constexpr int G = Red::kLanes, CAP = G * kCandPerLane;
            static_assert((64 / G) * CAP <= kCandCap, "the wave's slice holds every group's list");
            if (__ballot(cnt > try_below * ((float)CAP / (float)kCandCap)) == 0ull) {
                const int lane = threadIdx.x & 63, gl = lane % G;
                float* clist = cand + (lane / G) * CAP;
                int mine = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) mine += z[e] > tau ? 1 : 0;
                const int incl = group_scan_incl<G>(mine, gl);
                if (__ballot(incl > CAP) == 0ull) {  // (incl of a group's last lane is its total)
                    const int total = __shfl(incl, lane | (G - 1));
                    int off = incl - mine;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const bool in = z[e] > tau;
                        if (__ballot(in) != 0ull) {  // wave-uniform
                            if (in) clist[off] = z[e];
                            off += in ? 1 : 0;
                        }
                    }
                    wave_lds_sync();
#pragma unroll
                    for (int q = 0; q < kCandPerLane; ++q)
                        c[q] = (gl + G * q < total) ? clist[gl + G * q] : -INFINITY;
                    wave_lds_sync();
                    listed = true;
                    ++it;
                    break;
                }
                try_below *= 0.5f;
            }
        }
        if constexpr (NW >= 1) {
            // cnt bounds {z > tau_k} BEFORE the step; the set above the new tau is a subset
            if (cnt <= try_below) {                  // wave- (NW == 1) / workgroup-uniform
                const int lane = threadIdx.x & 63;
                if constexpr (NW == 1) {
                    int* ccols = list ? reinterpret_cast<int*>(cand + kCandCap) : nullptr;
                    const int n_c = compact_candidates<E, VEC>(z, tau, cand, kCandCap, ccols, VEC ? lane * 4 : lane,
                                                               VEC ? 256 : 64);
                    wave_lds_sync();
#pragma unroll
                    for (int q = 0; q < kCandPerLane; ++q) {
                        const bool have = lane + 64 * q < n_c;
                        c[q] = have ? cand[lane + 64 * q] : -INFINITY;
                        cc[q] = (have && list) ? ccols[lane + 64 * q] : 0;
                    }
                    wave_lds_sync();
                    listed = n_c <= kCandCap;
                } else {
                    const int wave = wave_index();
                    int* wcnt = reinterpret_cast<int*>(cand + NW * 64);
                    const int n_w = compact_candidates<E, VEC>(z, tau, cand + wave * 64, 64, nullptr, 0, 0);
                    if (lane == 0) wcnt[wave] = n_w;
                    __syncthreads();
                    int total = 0, worst = 0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        const int k = wcnt[w];
                        total += k;
                        worst = k > worst ? k : worst;
                    }
                    listed = worst <= 64 && total <= kCandCap;   // (a wave's list holds 64)
                    if (listed && wave == 0) {
#pragma unroll
                        for (int q = 0; q < kCandPerLane; ++q) {
                            const int idx = lane + 64 * q;
                            float v = -INFINITY;
                            int pre = 0;
#pragma unroll
                            for (int w = 0; w < NW; ++w) {
                                const int k = wcnt[w];
                                if (idx >= pre && idx < pre + k) v = cand[w * 64 + (idx - pre)];
                                pre += k;
                            }
                            c[q] = v;
                        }
                    }
                    __syncthreads();   // the lists are rewritten by the next attempt / the next row
                }
                if (listed) { ++it; break; }
                try_below = 0.5f * cnt;
            }
        }
    }
    if constexpr (NW >= 1) {
        if (listed) {
            // one wave iterates on the list (NW > 1: for the workgroup, which waits at one barrier;
            // run by every wave it cost a CU -- four waves to a SIMD -- four times the iteration)
            if (NW == 1 || wave_index() == 0) {
                WaveRed<64> wred;
                for (; it < kE15MaxNewton; ++it) {
                    float cnt;
                    const float step = entmax15_newton_step(c, tau, wred, cnt);
                    if (!(step > 1e-7f)) break;
                    tau += step;
                }
                tau = entmax15_closed_form(c, tau, wred);
            }
            if constexpr (NW > 1) {
                float* slot = cand + NW * 64 + NW;
                if (threadIdx.x == 0) *slot = tau;
                __syncthreads();
                tau = *slot;
            }
        }
    }
    if constexpr (NW == 0 && GC) {
        if (listed) {
            for (; it < kE15MaxNewton; ++it) {
                float cnt;
                const float step = entmax15_newton_step(c, tau, red, cnt);
                const bool moving = step > 1e-7f;
                if (moving) tau += step;
                if (__ballot(moving) == 0ull) break;
            }
            tau = entmax15_closed_form(c, tau, red);
        }
    }
    if (!listed) tau = entmax15_closed_form(z, tau, red);
    if constexpr (NW == 1) {
        if (list) {
            const int lane = threadIdx.x & 63;
            if (listed) emit_cols_from_candidates<kCandPerLane, 64>(c, cc, tau, lane, 0, list, cap, live, kColsOverflow, vdelta);
            else if (live && lane == 0) list[0] = kColsOverflow;
        }
    }
    int n = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const float d = fmaxf(z[e] - tau, 0.0f);
        z[e] = d * d;
        n += d > 0.0f ? 1 : 0;
    }
    nnz_out = red.sum(n);
    tau_out = tau;
}

// dx = sqrt(p) (dy - q), q = sum dy sqrt(p) / sum sqrt(p)   (Entmax15Function.backward)
template <int E, class Red>
__device__ __forceinline__ void entmax15_bwd_row(float (&g)[E], const float (&y)[E], Red& red)
{
    float a = 0.0f, b = 0.0f;
    float r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        r[e] = sqrtf(y[e]);
        a = fmaf(g[e], r[e], a);
        b += r[e];
    }
    const double A = red.sum((double)a), B = red.sum((double)b);
    const float q = (float)(A / B);
#pragma unroll
    for (int e = 0; e < E; ++e) g[e] = (g[e] - q) * r[e];
}

template <int G, int E, bool VEC, bool FULL>
__global__ __launch_bounds__(64 * kWavesPerBlock, E <= 32 ? 5 : 4) void entmax15_fwd_wave_kernel(
    const float* __restrict__ x, float* __restrict__ p, int32_t* __restrict__ nnz,
    float* __restrict__ tau, int64_t R, int K, int32_t* __restrict__ cols = nullptr, int cap = 0,
    long long vdelta = 0)
{
    const int lane = threadIdx.x & 63, gl = lane % G;
    int64_t row = wave_row<G>(lane);
    const bool live = row < R;
    if (!live) row = R - 1;
    float z[E];
    load_row<E, VEC, FULL>(x + row * K, K, gl, G, z, -INFINITY);
    WaveRed<G> red;
    int n;
    float t;
    constexpr bool CAND = (G == 64);   // one row per wave: a candidate list (values, columns) per wave
    constexpr bool GCAND = (G < 64) && (E >= 8);   // several rows per wave: a list of values per lane group
    __shared__ float cand[CAND ? kWavesPerBlock * 2 * kCandCap : GCAND ? kWavesPerBlock * kCandCap : 1];
    if constexpr (CAND)
        entmax15_row<E, 1, VEC>(z, red, n, t, cand + (threadIdx.x >> 6) * 2 * kCandCap,
                                cols ? cols + row * cap : nullptr, cap, live, vdelta);
    else if constexpr (GCAND)
        entmax15_row<E, 0, VEC, true>(z, red, n, t, cand + (threadIdx.x >> 6) * kCandCap);
    else
        entmax15_row<E, 0, VEC>(z, red, n, t);
    if (live) {
        store_row<E, VEC, FULL>(p + row * K, K, gl, G, z);
        if (gl == 0) {
            if (nnz) nnz[row] = n;
            if (tau) tau[row] = t;
        }
    }
}

template <int NW, int E, bool VEC, bool FULL>
__global__ __launch_bounds__(64 * NW, 4) void entmax15_fwd_block_kernel(
    const float* __restrict__ x, float* __restrict__ p, int32_t* __restrict__ nnz,
    float* __restrict__ tau, int64_t R, int K)
{
    __shared__ unsigned long long lds[BlockRed<NW>::kLdsWords];
    __shared__ float cand[NW * 64 + NW + 4];   // per-wave lists, counts, wave 0's tau
    for (int64_t row0 = blockIdx.x; row0 < R; row0 += gridDim.x) {
        const int64_t row = rot8(row0, R);
        float z[E];
        load_row<E, VEC, FULL, true>(x + row * K, K, threadIdx.x, 64 * NW, z, -INFINITY);
        BlockRed<NW> red(lds);
        int n;
        float t;
        entmax15_row<E, NW, VEC>(z, red, n, t, cand);
        store_row<E, VEC, FULL, true>(p + row * K, K, threadIdx.x, 64 * NW, z);
        if (threadIdx.x == 0) {
            if (nnz) nnz[row] = n;
            if (tau) tau[row] = t;
        }
        __syncthreads();
    }
}

template <int G, int E, bool VEC, bool FULL>
__global__ __launch_bounds__(64 * kWavesPerBlock, E <= 32 ? 4 : 2) void entmax15_bwd_wave_kernel(
    const float* __restrict__ dy, const float* __restrict__ p, float* __restrict__ dx, int64_t R, int K)
{
    const int lane = threadIdx.x & 63, gl = lane % G;
    int64_t row = wave_row<G>(lane);
    const bool live = row < R;
    if (!live) row = R - 1;
    float g[E], y[E];
    load_row<E, VEC, FULL>(p + row * K, K, gl, G, y, 0.0f);
    load_row<E, VEC, FULL>(dy + row * K, K, gl, G, g, 0.0f);
    WaveRed<G> red;
    entmax15_bwd_row(g, y, red);
    if (live) store_row<E, VEC, FULL>(dx + row * K, K, gl, G, g);
}

// Rows of one wave (512 < K <= 4096) with the forward's lists: columns of the support and the
// weights r = sqrt(p) = z - tau.  dx = r (dy - q) at the listed columns, q = sum dy r / sum r, zero
// elsewhere: the row of dy is gathered at <= cap places, p is not read, dx is zero-filled and the
// few values follow once the fill is acknowledged (the same two rounds of stores as sparsemax's
// first list backward).  Rows marked kColsOverflow read p and dy as entmax15_bwd_wave_kernel does.
template <int E, bool VEC, bool FULL>
__global__ __launch_bounds__(64 * kWavesPerBlock, E <= 32 ? 4 : 2) void entmax15_bwd_cols_wave_kernel(
    const float* __restrict__ dy, const float* __restrict__ p, const int32_t* __restrict__ cols,
    const float* __restrict__ rvals, int cap, float* __restrict__ dx, int64_t R, int K)
{
    const int lane = threadIdx.x & 63;
    int64_t row = wave_row<64>(lane);
    const bool live = row < R;
    if (!live) row = R - 1;
    WaveRed<64> red;
    const int c = lane < cap ? cols[row * cap + lane] : kColsNone;
    if (__builtin_amdgcn_readfirstlane(c) < kColsNone) {   // no list: the p-based backward
        float g[E], y[E];
        load_row<E, VEC, FULL>(p + row * K, K, lane, 64, y, 0.0f);
        load_row<E, VEC, FULL>(dy + row * K, K, lane, 64, g, 0.0f);
        entmax15_bwd_row(g, y, red);
        if (live) store_row<E, VEC, FULL>(dx + row * K, K, lane, 64, g);
        return;
    }
    float* dxrow = dx + row * K;
    if constexpr (VEC) {
        // 16-byte rows: dx assembled 1024 columns at a time in the wave's LDS tile and written once
        // with full-width stores, like sparsemax's list backward (zero fill, wait, scattered
        // 4-byte stores: 4.43 ms at 1M x 4096 against the p-based 8.56 ms; assembled: see DESIGN)
        __shared__ __align__(16) float pat_lds[kWavesPerBlock * kPatLdsWords];
        float* wl = pat_lds + (threadIdx.x >> 6) * kPatLdsWords;
        float4* tile4 = reinterpret_cast<float4*>(wl);
#pragma unroll
        for (int i = 0; i < kPatTile / 256; ++i) tile4[i * 64 + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float w = c >= 0 ? rvals[row * cap + lane] : 0.0f;
        bwd_row_assemble_weighted<E>(dy + row * K, dxrow, c, w, K, lane, live, red, wl);
        return;
    }
    if (live) {
        float zero[E];
#pragma unroll
        for (int e = 0; e < E; ++e) zero[e] = 0.0f;
        store_row<E, VEC, FULL>(dxrow, K, lane, 64, zero);
    }
    float r = 0.0f, g = 0.0f;
    if (c >= 0) {
        r = rvals[row * cap + lane];
        g = dy[row * K + c];
    }
    const double A = red.sum((double)(g * r)), B = red.sum((double)r);
    const float q = (float)(A / B);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (live && c >= 0) dxrow[c] = (g - q) * r;
}

template <int NW, int E, bool VEC, bool FULL>
__global__ __launch_bounds__(64 * NW) void entmax15_bwd_block_kernel(
    const float* __restrict__ dy, const float* __restrict__ p, float* __restrict__ dx, int64_t R, int K)
{
    __shared__ unsigned long long lds[BlockRed<NW>::kLdsWords];
    for (int64_t row0 = blockIdx.x; row0 < R; row0 += gridDim.x) {
        const int64_t row = rot8(row0, R);
        float g[E], y[E];
        load_row<E, VEC, FULL, true>(p + row * K, K, threadIdx.x, 64 * NW, y, 0.0f);
        load_row<E, VEC, FULL, true>(dy + row * K, K, threadIdx.x, 64 * NW, g, 0.0f);
        BlockRed<NW> red(lds);
        entmax15_bwd_row(g, y, red);
        store_row<E, VEC, FULL, true>(dx + row * K, K, threadIdx.x, 64 * NW, g);
        __syncthreads();
    }
}

struct E15Args {
    const float* in0;     // x (forward) / dy (backward)
    const float* in1;     // p (backward)
    float* out;           // p (forward) / dx (backward)
    int32_t* nnz;
    float* tau;
    int64_t R;
    int K;
    hipStream_t st;
    bool vec;
    int32_t* cols = nullptr;       // forward: support lists to write / backward: to read
    const float* rvals = nullptr;  // backward: sqrt(p) of the listed columns
    int cap = 0;
    long long vdelta = 0;          // forward: byte distance from the column list to the value list
};

#define SMLVM_E15_MODES(GO, A, E)                                                \
    do {                                                                         \
        const bool full = a.vec && a.K == A * E;                                 \
        if (!a.vec) GO<A, E, false, false>(a);                                   \
        else if (full) GO<A, E, true, true>(a);                                  \
        else GO<A, E, true, false>(a);                                           \
    } while (0)

template <int G, int E, bool VEC, bool FULL>
static void e15_fwd_wave_go(const E15Args& a)
{
    hipLaunchKernelGGL((entmax15_fwd_wave_kernel<G, E, VEC, FULL>), dim3(wave_grid(a.R, G)),
                       dim3(64 * kWavesPerBlock), 0, a.st, a.in0, a.out, a.nnz, a.tau, a.R, a.K,
                       G == 64 ? a.cols : nullptr, a.cap, a.vdelta);
}
template <int G, int E, bool VEC, bool FULL>
static void e15_bwd_cols_go(const E15Args& a)
{
    if constexpr (G == 64)
        hipLaunchKernelGGL((entmax15_bwd_cols_wave_kernel<E, VEC, FULL>), dim3(wave_grid(a.R, 64)),
                           dim3(64 * kWavesPerBlock), 0, a.st, a.in0, a.in1, a.cols, a.rvals, a.cap, a.out,
                           a.R, a.K);
}
template <int NW, int E, bool VEC, bool FULL>
static void e15_bwd_cols_block_go(const E15Args&) {}   // (never dispatched: lists are for one-wave rows)
template <int NW, int E, bool VEC, bool FULL>
static void e15_fwd_block_go(const E15Args& a)
{
    hipLaunchKernelGGL((entmax15_fwd_block_kernel<NW, E, VEC, FULL>), dim3(block_grid(a.R)), dim3(64 * NW),
                       0, a.st, a.in0, a.out, a.nnz, a.tau, a.R, a.K);
}
template <int G, int E, bool VEC, bool FULL>
static void e15_bwd_wave_go(const E15Args& a)
{
    hipLaunchKernelGGL((entmax15_bwd_wave_kernel<G, E, VEC, FULL>), dim3(wave_grid(a.R, G)),
                       dim3(64 * kWavesPerBlock), 0, a.st, a.in0, a.in1, a.out, a.R, a.K);
}
template <int NW, int E, bool VEC, bool FULL>
static void e15_bwd_block_go(const E15Args& a)
{
    hipLaunchKernelGGL((entmax15_bwd_block_kernel<NW, E, VEC, FULL>), dim3(block_grid(a.R)), dim3(64 * NW),
                       0, a.st, a.in0, a.in1, a.out, a.R, a.K);
}
template <int A, int E> static void launch_e15_fwd_wave(const E15Args& a) { SMLVM_E15_MODES(e15_fwd_wave_go, A, E); }
template <int A, int E> static void launch_e15_fwd_block(const E15Args& a) { SMLVM_E15_MODES(e15_fwd_block_go, A, E); }
template <int A, int E> static void launch_e15_bwd_wave(const E15Args& a) { SMLVM_E15_MODES(e15_bwd_wave_go, A, E); }
template <int A, int E> static void launch_e15_bwd_block(const E15Args& a) { SMLVM_E15_MODES(e15_bwd_block_go, A, E); }
template <int A, int E> static void launch_e15_bwd_cols(const E15Args& a) { SMLVM_E15_MODES(e15_bwd_cols_go, A, E); }
template <int A, int E> static void launch_e15_bwd_cols_block(const E15Args& a) { SMLVM_E15_MODES(e15_bwd_cols_block_go, A, E); }

}  // namespace smlvm

using namespace smlvm;

extern "C" int smlvm_sparsemax_fwd(const float* x, float* p, int32_t* supp_size, int32_t* nnz,
                                   float* tau, float* entropy, int64_t R, int32_t K, void* stream)
{
    if (R < 0 || K < 1) return SMLVM_ERR_INVALID;
    if (R == 0) return SMLVM_OK;
    if (!x || !p || !supp_size) return SMLVM_ERR_INVALID;
    FwdArgs a{x, p, supp_size, nnz, tau, entropy, R, K, (hipStream_t)stream,
              (K % 4 == 0) && aligned16(x) && aligned16(p), nnz || tau || entropy, nullptr, 0, nullptr};
    if (K > 65536) launch_fwd_stream(a);
    else SMLVM_DISPATCH(launch_fwd_wave, launch_fwd_block, a);
    return check_launch();
}

extern "C" size_t smlvm_sparsemax_support_mask_bytes(int64_t R, int32_t K)
{
    if (R <= 0 || K <= 32 || K > 4096) return 0;
    if (K <= 1024) {   // 16-register tiles: 16 bits per lane of the row's lane group (SMLVM_DISPATCH)
        const int lanes = K <= 128 ? 8 : (K <= 256 ? 16 : (K <= 512 ? 32 : 64));
        return (size_t)R * lanes * sizeof(uint16_t);
    }
    return (size_t)R * kMaskWordsPerRow * sizeof(uint32_t);
}

extern "C" int smlvm_sparsemax_fwd_cols(const float* x, float* p, int32_t* supp_size, int32_t* nnz,
                                        float* tau, float* entropy, int32_t* support_cols,
                                        int32_t cap, uint32_t* support_mask, int64_t R, int32_t K,
                                        void* stream)
{
    if (R < 0 || K < 1) return SMLVM_ERR_INVALID;
    if (support_cols && (cap < 1 || cap > (K > 4096 ? kColsMaxCapWide : kColsMaxCap))) return SMLVM_ERR_INVALID;
    if (R == 0) return SMLVM_OK;
    if (!x || !p || !supp_size) return SMLVM_ERR_INVALID;
    FwdArgs a{x, p, supp_size, nnz, tau, entropy, R, K, (hipStream_t)stream,
              (K % 4 == 0) && aligned16(x) && aligned16(p), nnz || tau || entropy, support_cols, cap,
              (support_cols && K > 32 && K <= 4096) ? support_mask : nullptr};
    if (K > 65536) launch_fwd_stream(a);
    else SMLVM_DISPATCH(launch_fwd_wave, launch_fwd_block, a);
    return check_launch();
}

extern "C" int smlvm_sparsemax_fwd_lists(const float* x, float* p, int32_t* supp_size, int32_t* nnz,
                                         float* tau, float* entropy, int32_t* support_cols,
                                         float* support_vals, int32_t cap, uint32_t* support_mask,
                                         int64_t R, int32_t K, void* stream)
{
    if (!support_vals)
        return smlvm_sparsemax_fwd_cols(x, p, supp_size, nnz, tau, entropy, support_cols, cap, support_mask, R, K, stream);
    if (R < 0 || K < 1 || !support_cols) return SMLVM_ERR_INVALID;
    if (K > 4096) return SMLVM_ERR_UNSUPPORTED;   // value lists: the one-wave-per-row kernels only
    if (cap < 1 || cap > kColsMaxCap) return SMLVM_ERR_INVALID;
    if (R == 0) return SMLVM_OK;
    if (!x || !p || !supp_size) return SMLVM_ERR_INVALID;
    FwdArgs a{x, p, supp_size, nnz, tau, entropy, R, K, (hipStream_t)stream,
              (K % 4 == 0) && aligned16(x) && aligned16(p), nnz || tau || entropy, support_cols, cap,
              K > 32 ? support_mask : nullptr, support_vals};
    SMLVM_DISPATCH(launch_fwd_wave, launch_fwd_block, a);
    return check_launch();
}

extern "C" int smlvm_sparsemax_bwd(const float* dy, const float* p, const int32_t* supp_size,
                                   float* dx, int64_t R, int32_t K, void* stream)
{
    if (R < 0 || K < 1) return SMLVM_ERR_INVALID;
    if (R == 0) return SMLVM_OK;
    if (!dy || !p || !supp_size || !dx) return SMLVM_ERR_INVALID;
    BwdArgs a{dy, p, supp_size, dx, R, K, (hipStream_t)stream,
              (K % 4 == 0) && aligned16(dy) && aligned16(p) && aligned16(dx), nullptr, 0, nullptr};
    if (K > 65536) launch_bwd_stream(a);
    else SMLVM_DISPATCH(launch_bwd_wave, launch_bwd_block, a);
    return check_launch();
}

extern "C" int smlvm_sparsemax_bwd_cols(const float* dy, const int32_t* support_cols, int32_t cap,
                                        const uint32_t* support_mask, const float* p,
                                        const int32_t* supp_size, float* dx, int64_t R, int32_t K,
                                        void* stream)
{
    if (R < 0 || K < 1) return SMLVM_ERR_INVALID;
    if (support_cols && (cap < 1 || cap > (K > 4096 ? kColsMaxCapWide : kColsMaxCap))) return SMLVM_ERR_INVALID;
    if (R == 0) return SMLVM_OK;
    if (!dy || !p || !supp_size || !dx) return SMLVM_ERR_INVALID;
    BwdArgs a{dy, p, supp_size, dx, R, K, (hipStream_t)stream,
              (K % 4 == 0) && aligned16(dy) && aligned16(p) && aligned16(dx),
              support_cols, cap, (support_cols && K > 32 && K <= 4096) ? support_mask : nullptr};
    if (K > 65536) launch_bwd_stream(a);
    else SMLVM_DISPATCH(launch_bwd_wave, launch_bwd_block, a);
    return check_launch();
}

extern "C" int smlvm_entropy(const float* p, float* ent, int64_t R, int32_t K, void* stream)
{
    if (R < 0 || K < 1) return SMLVM_ERR_INVALID;
    if (R == 0) return SMLVM_OK;
    if (!p || !ent) return SMLVM_ERR_INVALID;
    hipLaunchKernelGGL(entropy_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, p, ent, R, K);
    return check_launch();
}

extern "C" int smlvm_entmax15_fwd(const float* x, float* p, int32_t* nnz, float* tau, int64_t R,
                                  int32_t K, void* stream)
{
    if (R < 0 || K < 1) return SMLVM_ERR_INVALID;
    if (K > 65536) return SMLVM_ERR_UNSUPPORTED;
    if (R == 0) return SMLVM_OK;
    if (!x || !p) return SMLVM_ERR_INVALID;
    E15Args a{x, nullptr, p, nnz, tau, R, K, (hipStream_t)stream, (K % 4 == 0) && aligned16(x) && aligned16(p)};
    SMLVM_DISPATCH(launch_e15_fwd_wave, launch_e15_fwd_block, a);
    return check_launch();
}

// Rows held by one wave (512 < K <= 4096): the forward also writes each row's support list and the
// weights sqrt(p) of its entries; the backward then gathers dy at those columns and never reads p.
extern "C" int smlvm_entmax15_fwd_lists(const float* x, float* p, int32_t* nnz, float* tau,
                                        int32_t* support_cols, float* support_vals, int32_t cap,
                                        int64_t R, int32_t K, void* stream)
{
    if (R < 0 || K < 1 || cap < 1 || cap > kColsMaxCap) return SMLVM_ERR_INVALID;
    if (K <= 512 || K > 4096) return SMLVM_ERR_UNSUPPORTED;
    if (R == 0) return SMLVM_OK;
    if (!x || !p || !support_cols || !support_vals) return SMLVM_ERR_INVALID;
    E15Args a{x, nullptr, p, nnz, tau, R, K, (hipStream_t)stream, (K % 4 == 0) && aligned16(x) && aligned16(p)};
    a.cols = support_cols;
    a.cap = cap;
    a.vdelta = (long long)(reinterpret_cast<const char*>(support_vals) - reinterpret_cast<const char*>(support_cols));
    SMLVM_DISPATCH(launch_e15_fwd_wave, launch_e15_fwd_block, a);
    return check_launch();
}

extern "C" int smlvm_entmax15_bwd_lists(const float* dy, const float* p, const int32_t* support_cols,
                                        const float* support_vals, int32_t cap, float* dx, int64_t R,
                                        int32_t K, void* stream)
{
    if (R < 0 || K < 1 || cap < 1 || cap > kColsMaxCap) return SMLVM_ERR_INVALID;
    if (K <= 512 || K > 4096) return SMLVM_ERR_UNSUPPORTED;
    if (R == 0) return SMLVM_OK;
    if (!dy || !p || !dx || !support_cols || !support_vals) return SMLVM_ERR_INVALID;
    E15Args a{dy, p, dx, nullptr, nullptr, R, K, (hipStream_t)stream,
              (K % 4 == 0) && aligned16(dy) && aligned16(p) && aligned16(dx)};
    a.cols = const_cast<int32_t*>(support_cols);
    a.rvals = support_vals;
    a.cap = cap;
    SMLVM_DISPATCH(launch_e15_bwd_cols, launch_e15_bwd_cols_block, a);
    return check_launch();
}

extern "C" int smlvm_entmax15_bwd(const float* dy, const float* p, float* dx, int64_t R, int32_t K,
                                  void* stream)
{
    if (R < 0 || K < 1) return SMLVM_ERR_INVALID;
    if (K > 65536) return SMLVM_ERR_UNSUPPORTED;
    if (R == 0) return SMLVM_OK;
    if (!dy || !p || !dx) return SMLVM_ERR_INVALID;
    E15Args a{dy, p, dx, nullptr, nullptr, R, K, (hipStream_t)stream,
              (K % 4 == 0) && aligned16(dy) && aligned16(p) && aligned16(dx)};
    SMLVM_DISPATCH(launch_e15_bwd_wave, launch_e15_bwd_block, a);
    return check_launch();
}
