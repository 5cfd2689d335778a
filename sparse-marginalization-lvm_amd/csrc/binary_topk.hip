// k-best assignments of n independent binary variables, batched on the GPU.
//
// Replaces lvmhelpers/pbinary_topk.pyx:23-34 (batched_topk) and the per-row host DP of
// lvmhelpers/binary_topk.h:57-129.  One group of G lanes (16 or 64) per sample, list slot j on lane
// j (k <= 64; up to k = 1024 one workgroup per sample, see binary_topk_block_kernel).  Each DP step merges the sorted list `a` (bit i off) with `b = a + x_i` (bit i on) by merge
// ranks instead of a serial two-pointer walk:
//     rank(a_j) = j + #{l : b_l >  a_j}      (b is taken only if STRICTLY greater, binary_topk.h:72)
//     rank(b_l) = l + #{j : a_j >= b_l}
// which reproduces the reference's merge order exactly, ties included; ranks >= k are dropped.
// Scores are fp32 sums accumulated in index order (binary_topk.h:28), so they match bit for bit.
#include "common.hpp"

#include <math.h>

namespace smlvm {

// LDS traffic of one lane group stays inside one wave (G divides 64), and a wave's LDS operations
// execute in order: this only pins the compiler's order and re-converges the wave.
__device__ __forceinline__ void group_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// lane T of every 16-lane DPP row, broadcast to the whole row (DPP row_newbcast, gfx90a and later)
template <int T>
__device__ __forceinline__ float row_bcast(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v),
                                                                 0x150 + T, 0xf, 0xf, false));
}

template <int G, int W>
__global__ __launch_bounds__(256) void binary_topk_kernel(const float* __restrict__ scores,
                                                          float* __restrict__ configs,
                                                          float* __restrict__ topscores, int64_t B,
                                                          int n, int k)
{
    constexpr int GPB = 256 / G;  // groups (samples) per block
    __shared__ float s_score[256];
    __shared__ int s_src[256];
    __shared__ uint32_t s_cfg[256 * W];
    // the sample's scores, staged once: the DP below is n dependent steps, and a global load per
    // step (~1 us each) was most of its time
    __shared__ float s_x[GPB * 32 * W];

    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int64_t sample = (int64_t)blockIdx.x * GPB + grp;
    const bool live = sample < B;
    const float* x = scores + (live ? sample : B - 1) * n;
    float* gs = s_score + grp * G;
    int* gi = s_src + grp * G;

    uint32_t cfg[W];
#pragma unroll
    for (int w = 0; w < W; ++w) cfg[w] = 0u;
    float* sx = s_x + grp * 32 * W;
    for (int i = lane; i < n; i += G) sx[i] = x[i];
    group_lds_sync();
    const float x0 = sx[0];
    const float c1 = 0.0f + x0;  // c0.update(0, x[0]), binary_topk.h:104-107
    const bool on_first = x0 >= 0.0f;  // binary_topk.h:110-116
    float a = -INFINITY;
    if (lane == 0) { a = on_first ? c1 : 0.0f; cfg[0] = on_first ? 1u : 0u; }
    if (lane == 1) { a = on_first ? 0.0f : c1; cfg[0] = on_first ? 0u : 1u; }
    int L = 2;

    for (int i = 1; i < n; ++i) {
        const float xi = sx[i];
        const float b = a + xi;
        int cntA = 0, cntB = 0;
        if constexpr (G == 16) {
            // a lane group is one DPP row: entry t reaches the other 15 lanes with row_newbcast
            // (one VALU instruction) instead of a ds_bpermute round trip through the LDS
#define SMLVM_TOPK_STEP(T)                                                    \
            if (T < L) {                                                      \
                const float at = row_bcast<T>(a);                             \
                const float bt = at + xi;                                     \
                cntA += (bt > a) ? 1 : 0;                                     \
                cntB += (at >= b) ? 1 : 0;                                    \
            }
            SMLVM_TOPK_STEP(0) SMLVM_TOPK_STEP(1) SMLVM_TOPK_STEP(2) SMLVM_TOPK_STEP(3)
            SMLVM_TOPK_STEP(4) SMLVM_TOPK_STEP(5) SMLVM_TOPK_STEP(6) SMLVM_TOPK_STEP(7)
            SMLVM_TOPK_STEP(8) SMLVM_TOPK_STEP(9) SMLVM_TOPK_STEP(10) SMLVM_TOPK_STEP(11)
            SMLVM_TOPK_STEP(12) SMLVM_TOPK_STEP(13) SMLVM_TOPK_STEP(14) SMLVM_TOPK_STEP(15)
#undef SMLVM_TOPK_STEP
        } else {
            for (int t = 0; t < L; ++t) {
                const float at = __shfl(a, t, G);
                const float bt = at + xi;
                cntA += (bt > a) ? 1 : 0;
                cntB += (at >= b) ? 1 : 0;
            }
        }
        const int rA = lane + cntA, rB = lane + cntB;
        if (lane < L && rA < k) { gs[rA] = a; gi[rA] = lane; }
        if (lane < L && rB < k) { gs[rB] = b; gi[rB] = lane | 0x100; }
        group_lds_sync();   // a group never spans waves: no workgroup barrier in the n-step loop
        const int Ln = (2 * L < k) ? 2 * L : k;
        float na = -INFINITY;
        int src = lane;
        if (lane < Ln) { na = gs[lane]; src = gi[lane]; }
        group_lds_sync();
        const int sl = src & 0xff;
        const uint32_t bit = (src & 0x100) ? (1u << (i & 31)) : 0u;
        const int wi = i >> 5;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            uint32_t v = (uint32_t)__shfl((int)cfg[w], sl, G);
            cfg[w] = v | ((w == wi) ? bit : 0u);
        }
        a = na;
        L = Ln;
    }

    // cooperative, coalesced write of the [k, n] block of 0/1 floats
    uint32_t* gc = s_cfg + grp * G * W;
#pragma unroll
    for (int w = 0; w < W; ++w) gc[lane * W + w] = cfg[w];
    group_lds_sync();
    if (live) {
        float* out = configs + sample * (int64_t)k * n;
        const int total = k * n;
        for (int idx = lane; idx < total; idx += G) {
            const int j = idx / n, i = idx - j * n;
            out[idx] = (float)((gc[j * W + (i >> 5)] >> (i & 31)) & 1u);
        }
        if (topscores && lane < k) topscores[sample * k + lane] = a;
    }
}

// k > 64: the list no longer fits a wave.  One workgroup of T threads per sample, slot j on thread j,
// everything that crossed lanes above goes through LDS.  With up to 1024 entries a linear count per
// slot would be O(k) LDS reads per step; both lists are sorted (a by the DP's invariant, b = a + x_i
// because fp32 addition of one x_i is monotone), so the two counts are the boundaries of monotone
// predicates -- evaluated with the SAME fp32 expressions as the linear count -- and a binary search
// finds each in <= 11 reads.  Same ranks, same tie rule, same bits as the wave kernel.
template <int T, int W>
__global__ __launch_bounds__(T) void binary_topk_block_kernel(const float* __restrict__ scores,
                                                              float* __restrict__ configs,
                                                              float* __restrict__ topscores,
                                                              int64_t B, int n, int k)
{
    __shared__ float s_a[T];
    __shared__ float s_score[T];
    __shared__ int s_src[T];
    __shared__ uint32_t s_cfg[W * T];   // word w of slot j at [w * T + j]
    __shared__ float s_x[32 * W];

    const int j = threadIdx.x;
    const int64_t sample = blockIdx.x;   // grid == B
    const float* x = scores + sample * n;
    uint32_t cfg[W];
#pragma unroll
    for (int w = 0; w < W; ++w) cfg[w] = 0u;
    for (int i = j; i < n; i += T) s_x[i] = x[i];
    __syncthreads();
    const float x0 = s_x[0];
    const float c1 = 0.0f + x0;
    const bool on_first = x0 >= 0.0f;
    float a = -INFINITY;
    if (j == 0) { a = on_first ? c1 : 0.0f; cfg[0] = on_first ? 1u : 0u; }
    if (j == 1) { a = on_first ? 0.0f : c1; cfg[0] = on_first ? 0u : 1u; }
    int L = 2;

    for (int i = 1; i < n; ++i) {
        const float xi = s_x[i];
        const float b = a + xi;
        s_a[j] = a;
#pragma unroll
        for (int w = 0; w < W; ++w) s_cfg[w * T + j] = cfg[w];
        __syncthreads();
        if (j < L) {
            // cntA = #{t < L : a_t + x_i > a_j}, cntB = #{t < L : a_t >= b_j}: true ... true false ... false
            int lo = 0, hi = L;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const float bt = s_a[mid] + xi;
                if (bt > a) lo = mid + 1; else hi = mid;
            }
            const int rA = j + lo;
            lo = 0;
            hi = L;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_a[mid] >= b) lo = mid + 1; else hi = mid;
            }
            const int rB = j + lo;
            if (rA < k) { s_score[rA] = a; s_src[rA] = j; }
            if (rB < k) { s_score[rB] = b; s_src[rB] = j | 0x10000; }
        }
        __syncthreads();
        const int Ln = (2 * L < k) ? 2 * L : k;
        float na = -INFINITY;
        int src = j;
        if (j < Ln) { na = s_score[j]; src = s_src[j]; }
        const int sl = src & 0xffff;
        const uint32_t bit = (src & 0x10000) ? (1u << (i & 31)) : 0u;
        const int wi = i >> 5;
#pragma unroll
        for (int w = 0; w < W; ++w) cfg[w] = s_cfg[w * T + sl] | ((w == wi) ? bit : 0u);
        __syncthreads();   // s_a / s_cfg / s_score are rewritten by the next step
        a = na;
        L = Ln;
    }

#pragma unroll
    for (int w = 0; w < W; ++w) s_cfg[w * T + j] = cfg[w];
    __syncthreads();
    float* out = configs + sample * (int64_t)k * n;
    const int total = k * n;
    for (int idx = j; idx < total; idx += T) {
        const int jj = idx / n, i = idx - jj * n;
        out[idx] = (float)((s_cfg[(i >> 5) * T + jj] >> (i & 31)) & 1u);
    }
    if (topscores && j < k) topscores[sample * k + j] = a;
}

template <int T>
static void launch_topk_block(const float* scores, float* configs, float* topscores, int64_t B,
                              int n, int k, hipStream_t st)
{
    const int W = (n + 31) / 32;
#define SMLVM_TOPK_GO(WW)                                                                        \
    hipLaunchKernelGGL((binary_topk_block_kernel<T, WW>), dim3((unsigned)B), dim3(T), 0, st, scores, \
                       configs, topscores, B, n, k)
    if (W <= 1) SMLVM_TOPK_GO(1);
    else if (W <= 2) SMLVM_TOPK_GO(2);
    else if (W <= 4) SMLVM_TOPK_GO(4);
    else if (W <= 8) SMLVM_TOPK_GO(8);
    else SMLVM_TOPK_GO(16);
#undef SMLVM_TOPK_GO
}

template <int G>
static void launch_topk(const float* scores, float* configs, float* topscores, int64_t B, int n,
                        int k, hipStream_t st)
{
    const unsigned grid = (unsigned)((B + (256 / G) - 1) / (256 / G));
    const int W = (n + 31) / 32;
#define SMLVM_TOPK_GO(WW)                                                                         \
    hipLaunchKernelGGL((binary_topk_kernel<G, WW>), dim3(grid), dim3(256), 0, st, scores, configs, \
                       topscores, B, n, k)
    if (W <= 1) SMLVM_TOPK_GO(1);
    else if (W <= 2) SMLVM_TOPK_GO(2);
    else if (W <= 4) SMLVM_TOPK_GO(4);
    else if (W <= 8) SMLVM_TOPK_GO(8);
    else SMLVM_TOPK_GO(16);
#undef SMLVM_TOPK_GO
}

}  // namespace smlvm

using namespace smlvm;

extern "C" int smlvm_binary_topk_batched(const float* scores, float* configs, float* topscores,
                                         int64_t B, int32_t n, int32_t k, void* stream)
{
    if (B < 0 || n < 1 || k < 2) return SMLVM_ERR_INVALID;
    if (n < 31 && (int64_t)k > ((int64_t)1 << n)) return SMLVM_ERR_INVALID;
    if (n > SMLVM_TOPK_MAX_BITS || k > SMLVM_TOPK_MAX_K) return SMLVM_ERR_UNSUPPORTED;
    if (B == 0) return SMLVM_OK;
    if (!scores || !configs) return SMLVM_ERR_INVALID;
    if (B > 0x7fffffffll) return SMLVM_ERR_UNSUPPORTED;
    if (k <= 16) launch_topk<16>(scores, configs, topscores, B, n, k, (hipStream_t)stream);
    else if (k <= 64) launch_topk<64>(scores, configs, topscores, B, n, k, (hipStream_t)stream);
    else if (k <= 256) launch_topk_block<256>(scores, configs, topscores, B, n, k, (hipStream_t)stream);
    else launch_topk_block<1024>(scores, configs, topscores, B, n, k, (hipStream_t)stream);
    return check_launch();
}
