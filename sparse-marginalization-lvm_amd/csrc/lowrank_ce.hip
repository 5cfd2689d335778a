// Next row after the hot path: the decoder side of the bit-vector VAE over the
// compacted support.  The reference's CategoricalGenerator (experiments/bit_vector-vae/archs.py:18-29)
// materialises logits [N, 784, 256] = X[N,784,R] . W[256,R]^T + b (R = out_rank = 5) and
// reconstruction_loss (experiments/bit_vector-vae/train.py:361-371) runs a 256-way cross entropy
// over them, summed over the 784 features: 0.8 MB of logits per support row, written and re-read
// several times.  These kernels never materialise the logits: a logit costs R FMAs to recompute,
// so forward and backward recompute them from X (20 B per feature) and keep only the per-feature
// log-sum-exp.  Compute bound (fp32 VALU FMA + v_exp_f32); HBM traffic is ~32 B per (row, feature).
//
//   fwd   lane = two (row, feature) pairs: one exp pass over the C classes, shifted by an upper
//         bound of the logits (exact two-pass redo if that underflows); W and b are read with
//         wave-uniform addresses (scalar loads) and feed v_pk_fma_f32; then a per-row sum.
//   bwd   dX: lane = one (row, feature), softmax from the saved lse, dX = g (softmax - onehot) W
//         dW, db: thread = one class, loops over (row, feature) tiles staged in LDS; per-block
//         partials, then a deterministic fp64 reduction over blocks (no float atomics).
#include "common.hpp"

#include <math.h>

namespace smlvm {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 splat(float v) { f32x2 r = {v, v}; return r; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// raw v_exp_f32: arguments here are <= 0 (or bounded), results that underflow may flush to 0
__device__ __forceinline__ float ex2(float v) { return __builtin_amdgcn_exp2f(v); }

// Label of pair i, or -1 when it is outside [0, C): nothing is ever indexed with such a label (the
// reference's F.cross_entropy raises a device assert there, experiments/bit_vector-vae/train.py:368);
// the forward turns it into a NaN loss for that row, the backward into NaN dx, and the one-hot term
// of dW / db simply matches no class.
template <typename YT>
__device__ __forceinline__ int label_at(const void* y, int64_t i, int C)
{
    const long long v = (long long)static_cast<const YT*>(y)[i];
    return (v >= 0 && v < (long long)C) ? (int)v : -1;
}

// Every lane carries TWO (row, feature) pairs -- p and p + 256 of the block's 512 -- so that the
// R FMAs of a logit and the shift run as v_pk_fma_f32 with the (wave-uniform) weight in a scalar
// register; only the v_exp_f32 are per element.
template <int R>
__device__ __forceinline__ f32x2 logit2(const f32x2 (&x)[R], const float* __restrict__ w,
                                        const float* __restrict__ b, int c)
{
    f32x2 l = splat(b[c]);
#pragma unroll
    for (int r = 0; r < R; ++r) l = fma2(x[r], splat(w[c * R + r]), l);
    return l;
}

// logit of one (divergent) class for one element: plain loads
template <int R>
__device__ __forceinline__ float logit_at(const f32x2 (&x)[R], int half, const float* __restrict__ w,
                                          const float* __restrict__ b, int c)
{
    float l = b[c];
#pragma unroll
    for (int r = 0; r < R; ++r) l = fmaf(half ? x[r].y : x[r].x, w[c * R + r], l);
    return l;
}

// exact two-pass log-sum-exp of one element (only when the bound-shifted sum underflowed)
template <int R>
__device__ float lse_exact(const f32x2 (&x)[R], int half, const float* __restrict__ w,
                           const float* __restrict__ b, int C)
{
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, logit_at<R>(x, half, w, b, c));
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s += ex2((logit_at<R>(x, half, w, b, c) - mx) * kLog2e);
    return fmaf(log2f(s), kLn2, mx);
}

// sum_c softmax_c W[c, :] - W[y, :] of one element, exactly (two passes; the underflow redo of DXD)
template <int R>
__device__ void dxdir_exact(const f32x2 (&x)[R], int half, const float* __restrict__ w,
                            const float* __restrict__ b, int C, int yy, float* __restrict__ out)
{
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, logit_at<R>(x, half, w, b, c));
    float s = 0.0f, v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float e = ex2((logit_at<R>(x, half, w, b, c) - mx) * kLog2e);
        s += e;
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = fmaf(e, w[c * R + r], v[r]);
    }
    const float inv = 1.0f / s;
#pragma unroll
    for (int r = 0; r < R; ++r) out[r] = yy >= 0 ? v[r] * inv - w[yy * R + r] : __builtin_nanf("");
}

constexpr int kPixPerBlock = 512;

// DXD (round 3): the pass that sums exp(logit - bound) also sums exp(logit - bound) W[c, :], R more
// packed FMAs per class; divided by the first sum that is softmax . W, and
//     dxdir[p, :] = sum_c softmax_c W[c, :] - W[y_p, :]
// is the gradient of the element's loss with respect to x[p, :].  The backward's dX is then
// gloss[n] * dxdir -- the dX kernel and its whole exp pass (one of the step's three) are gone.
template <int R, typename YT, bool DXD>
__global__ __launch_bounds__(256) void lowrank_ce_fwd_kernel(
    const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ b,
    const void* __restrict__ y, float* __restrict__ lse, float* __restrict__ nll,
    float* __restrict__ dxdir, int64_t P, int C)
{
    // Upper bound of every logit of an element: max_c b_c + sum_r |x_r| max_c |W_cr|.  Shifting by
    // it (instead of by the exact maximum) saves the whole max pass; the sum cannot overflow, and
    // if the bound is so loose that it underflows the element is redone exactly.
    __shared__ float red[4][R + 1];
    {
        const int c = threadIdx.x;
        float v[R + 1];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = c < C ? fabsf(W[c * R + r]) : 0.0f;
        v[R] = c < C ? b[c] : -INFINITY;
#pragma unroll
        for (int r = 0; r <= R; ++r) v[r] = group_reduce<64>(v[r], OpMaxF{});
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int r = 0; r <= R; ++r) red[threadIdx.x >> 6][r] = v[r];
        }
    }
    __syncthreads();
    float wabs[R + 1];
#pragma unroll
    for (int r = 0; r <= R; ++r)
        wabs[r] = fmaxf(fmaxf(red[0][r], red[1][r]), fmaxf(red[2][r], red[3][r]));

    int64_t p[2];
    bool live[2];
    p[0] = (int64_t)blockIdx.x * kPixPerBlock + threadIdx.x;
    p[1] = p[0] + 256;
#pragma unroll
    for (int h = 0; h < 2; ++h) { live[h] = p[h] < P; if (!live[h]) p[h] = P - 1; }
    f32x2 x[R];
    f32x2 bound = splat(wabs[R]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        x[r].x = X[p[0] * R + r];
        x[r].y = X[p[1] * R + r];
        bound.x = fmaf(fabsf(x[r].x), wabs[r], bound.x);
        bound.y = fmaf(fabsf(x[r].y), wabs[r], bound.y);
    }
    const f32x2 m2 = bound * kLog2e;
    const f32x2 k2 = splat(kLog2e);
    f32x2 s = splat(0.0f);
    f32x2 V[DXD ? R : 1];
#pragma unroll
    for (int r = 0; r < (DXD ? R : 1); ++r) V[r] = splat(0.0f);
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const f32x2 t = fma2(logit2<R>(x, W, b, c), k2, -m2);
        f32x2 e = {ex2(t.x), ex2(t.y)};
        s += e;
        if constexpr (DXD) {
#pragma unroll
            for (int r = 0; r < R; ++r) V[r] = fma2(e, splat(W[c * R + r]), V[r]);
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float sh = h ? s.y : s.x, bh = h ? bound.y : bound.x;
        const bool redo = !(sh > 1e-18f);   // also catches NaN
        float L = fmaf(log2f(sh), kLn2, bh);
        if (redo) L = lse_exact<R>(x, h, W, b, C);
        if (live[h]) {
            const int yy = label_at<YT>(y, p[h], C);
            lse[p[h]] = L;
            nll[p[h]] = yy >= 0 ? L - logit_at<R>(x, h, W, b, yy) : __builtin_nanf("");
            if constexpr (DXD) {
                float* out = dxdir + p[h] * R;
                if (redo) {
                    dxdir_exact<R>(x, h, W, b, C, yy, out);
                } else {
                    const float inv = 1.0f / sh;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float v = h ? V[r].y : V[r].x;
                        out[r] = yy >= 0 ? v * inv - W[yy * R + r] : __builtin_nanf("");
                    }
                }
            }
        }
    }
}

// loss[n] = sum_f nll[n, f]: one wave per row, fp64 partials, fixed reduction tree
__global__ __launch_bounds__(256) void row_sum_kernel(const float* __restrict__ v,
                                                      float* __restrict__ out, int64_t N, int F)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;  // whole wave
    const float* r = v + row * F;
    double acc = 0.0;
    for (int f = lane; f < F; f += 64) acc += (double)r[f];
    acc = group_reduce<64>(acc, OpSum{});
    if (lane == 0) out[row] = (float)acc;
}

// dX[p, r] = g (sum_c softmax_c W[c, r] - W[y, r])
template <int R, typename YT>
__global__ __launch_bounds__(256) void lowrank_ce_bwd_dx_kernel(
    const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ b,
    const void* __restrict__ y, const float* __restrict__ lse, const float* __restrict__ gloss,
    float* __restrict__ dX, int64_t P, int F, int C)
{
    int64_t p[2];
    bool live[2];
    p[0] = (int64_t)blockIdx.x * kPixPerBlock + threadIdx.x;
    p[1] = p[0] + 256;
#pragma unroll
    for (int h = 0; h < 2; ++h) { live[h] = p[h] < P; if (!live[h]) p[h] = P - 1; }
    f32x2 x[R], dx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        x[r].x = X[p[0] * R + r];
        x[r].y = X[p[1] * R + r];
        dx[r] = splat(0.0f);
    }
    f32x2 L2 = {lse[p[0]], lse[p[1]]};
    L2 *= kLog2e;
    const f32x2 k2 = splat(kLog2e);
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const f32x2 t = fma2(logit2<R>(x, W, b, c), k2, -L2);
        const f32x2 pr = {ex2(t.x), ex2(t.y)};
#pragma unroll
        for (int r = 0; r < R; ++r) dx[r] = fma2(pr, splat(W[c * R + r]), dx[r]);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (!live[h]) continue;
        const int yy = label_at<YT>(y, p[h], C);
        const float g = gloss[p[h] / F];
#pragma unroll
        for (int r = 0; r < R; ++r)
            dX[p[h] * R + r] = yy >= 0 ? g * ((h ? dx[r].y : dx[r].x) - W[yy * R + r]) : __builtin_nanf("");
    }
}

constexpr int kDwTile = 256;      // (row, feature) pairs staged per step (even)
constexpr int kDwMaxBlocks = 2048;

// dW[c, r] = sum_p g_p (softmax_pc - [c == y_p]) x_pr, db[c] likewise with x = 1.  Lane l of every
// wave owns the classes l, l + 64, l + 128, l + 192; the four waves of a block split the (row,
// feature) pairs of a tile between them, two pairs per step from an LDS tile stored [r][pair] (a
// pair is one broadcast ds_read_b64).  With thread = one class and every wave reading every pair
// (the first version) the LDS return path was the bound -- 32 B x 64 lanes x 4 waves per pair;
// now each pair is read by one wave.  The waves' accumulators are folded through LDS in wave order.
constexpr int kDwCls = 4;   // classes per lane: C <= 256
template <int R, typename YT>
__global__ __launch_bounds__(256) void lowrank_ce_bwd_dw_kernel(
    const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ b,
    const void* __restrict__ y, const float* __restrict__ lse, const float* __restrict__ gloss,
    float* __restrict__ partial, int64_t P, int F, int C)
{
    __shared__ __align__(8) float xs[R * kDwTile];
    __shared__ __align__(8) float ls[kDwTile];
    __shared__ __align__(8) float gs[kDwTile];
    __shared__ __align__(8) int ys[kDwTile];
    __shared__ float fold[3][kDwCls * 64 * (R + 1)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float w[kDwCls][R], bc[kDwCls];
    f32x2 acc[kDwCls][R + 1];
#pragma unroll
    for (int k = 0; k < kDwCls; ++k) {
        const int c = lane + 64 * k;
        bc[k] = c < C ? b[c] : 0.0f;
#pragma unroll
        for (int r = 0; r < R; ++r) { w[k][r] = c < C ? W[c * R + r] : 0.0f; acc[k][r] = splat(0.0f); }
        acc[k][R] = splat(0.0f);
    }
    const f32x2 k2 = splat(kLog2e);
    constexpr int kPerWave = kDwTile / 4;
    static_assert(kDwTile == 256, "one pair of the tile per thread");
    const int64_t tiles = (P + kDwTile - 1) / kDwTile;
    // The next tile travels from HBM into registers while the current one is worked on from LDS
    // (two workgroups per CU do not cover a tile's load latency between two barriers).
    float nx[R], nl = 0.0f, ng = 0.0f;
    int ny = -1;
    auto fetch = [&](int64_t t) {
        const int64_t p0 = t * kDwTile;
        const int cnt = (int)((P - p0 < kDwTile) ? (P - p0) : kDwTile);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int i = threadIdx.x + 256 * j;
            nx[j] = (i / R) < cnt ? X[p0 * R + i] : 0.0f;
        }
        const bool ok = (int)threadIdx.x < cnt;
        const int64_t p = ok ? p0 + threadIdx.x : p0;
        nl = lse[p] * kLog2e;
        ng = ok ? gloss[p / F] : 0.0f;   // padding contributes nothing
        ny = ok ? label_at<YT>(y, p, C) : -1;
    };
    if ((int64_t)blockIdx.x < tiles) fetch(blockIdx.x);
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int i = threadIdx.x + 256 * j;
            const int pix = i / R, r = i - pix * R;
            xs[r * kDwTile + pix] = nx[j];
        }
        ls[threadIdx.x] = nl;
        gs[threadIdx.x] = ng;
        ys[threadIdx.x] = ny;
        __syncthreads();
        if (t + gridDim.x < tiles) fetch(t + gridDim.x);
#pragma unroll 2
        for (int ii = 0; ii < kPerWave; ii += 2) {
            const int i = wave * kPerWave + ii;
            f32x2 xr[R];
#pragma unroll
            for (int r = 0; r < R; ++r) xr[r] = *reinterpret_cast<const f32x2*>(&xs[r * kDwTile + i]);
            const f32x2 lz = *reinterpret_cast<const f32x2*>(&ls[i]);
            const f32x2 g = *reinterpret_cast<const f32x2*>(&gs[i]);
            const int y0 = ys[i], y1 = ys[i + 1];
#pragma unroll
            for (int k = 0; k < kDwCls; ++k) {
                const int c = lane + 64 * k;
                f32x2 l = splat(bc[k]);
#pragma unroll
                for (int r = 0; r < R; ++r) l = fma2(xr[r], splat(w[k][r]), l);
                const f32x2 tt = fma2(l, k2, -lz);
                const f32x2 pr = {ex2(tt.x), ex2(tt.y)};
                const f32x2 hot = {(c == y0) ? 1.0f : 0.0f, (c == y1) ? 1.0f : 0.0f};
                const f32x2 dl = g * (pr - hot);
#pragma unroll
                for (int r = 0; r < R; ++r) acc[k][r] = fma2(dl, xr[r], acc[k][r]);
                acc[k][R] += dl;
            }
        }
    }
    // fold the four waves: waves 1..3 park their sums, wave 0 adds them in wave order
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int k = 0; k < kDwCls; ++k)
#pragma unroll
            for (int r = 0; r <= R; ++r)
                fold[wave - 1][(k * (R + 1) + r) * 64 + lane] = acc[k][r].x + acc[k][r].y;
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < kDwCls; ++k) {
            const int c = lane + 64 * k;
            if (c < C) {
                float* o = partial + ((int64_t)blockIdx.x * C + c) * (R + 1);
#pragma unroll
                for (int r = 0; r <= R; ++r) {
                    float v = acc[k][r].x + acc[k][r].y;
#pragma unroll
                    for (int q = 0; q < 3; ++q) v += fold[q][(k * (R + 1) + r) * 64 + lane];
                    o[r] = v;
                }
            }
        }
    }
}

// dW[c, r] / db[c] = sum over blocks of the partials: one wave per output, lanes stride over the
// blocks in fp64, fixed DPP tree at the end (deterministic)
__global__ __launch_bounds__(256) void lowrank_ce_reduce_kernel(const float* __restrict__ partial,
                                                                int nblk, int C, int R,
                                                                float* __restrict__ dW,
                                                                float* __restrict__ db)
{
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int per = C * (R + 1);
    if (j >= per) return;  // whole wave
    double acc = 0.0;
    for (int k = lane; k < nblk; k += 64) acc += (double)partial[(int64_t)k * per + j];
    acc = group_reduce<64>(acc, OpSum{});
    if (lane != 0) return;
    const int c = j / (R + 1), r = j - c * (R + 1);
    if (r < R) { if (dW) dW[c * R + r] = (float)acc; }
    else if (db) db[c] = (float)acc;
}

static int dw_blocks(int64_t P)
{
    const int64_t tiles = (P + kDwTile - 1) / kDwTile;
    return (int)(tiles < kDwMaxBlocks ? tiles : kDwMaxBlocks);
}

}  // namespace smlvm

using namespace smlvm;

#define SMLVM_LR_DISPATCH_R(R_, CALL)       \
    switch (R_) {                           \
        case 1: { constexpr int RR = 1; CALL; } break; \
        case 2: { constexpr int RR = 2; CALL; } break; \
        case 3: { constexpr int RR = 3; CALL; } break; \
        case 4: { constexpr int RR = 4; CALL; } break; \
        case 5: { constexpr int RR = 5; CALL; } break; \
        case 6: { constexpr int RR = 6; CALL; } break; \
        case 7: { constexpr int RR = 7; CALL; } break; \
        case 8: { constexpr int RR = 8; CALL; } break; \
        default: return SMLVM_ERR_UNSUPPORTED;          \
    }
#define SMLVM_LR_DISPATCH_Y(YK, CALL)                            \
    switch (YK) {                                                \
        case 0: { using YY = uint8_t; CALL; } break;             \
        case 1: { using YY = int32_t; CALL; } break;             \
        case 2: { using YY = int64_t; CALL; } break;             \
        default: return SMLVM_ERR_INVALID;                       \
    }

static int lr_check(int64_t N, int32_t F, int32_t C, int32_t R)
{
    if (N < 0 || F < 1 || C < 1 || R < 1) return SMLVM_ERR_INVALID;
    if (R > 8 || C > 256 || N * (int64_t)F > ((int64_t)1 << 40)) return SMLVM_ERR_UNSUPPORTED;
    return SMLVM_OK;
}

static int lowrank_ce_fwd_go(const float* x, const float* w, const float* b, const void* y,
                             int32_t y_kind, float* lse, float* nll, float* loss, float* dxdir,
                             int64_t N, int32_t F, int32_t C, int32_t R, void* stream)
{
    const int rc = lr_check(N, F, C, R);
    if (rc != SMLVM_OK) return rc;
    if (N == 0) return SMLVM_OK;
    if (!x || !w || !b || !y || !lse || !nll) return SMLVM_ERR_INVALID;
    const int64_t P = N * F;
    const unsigned grid = (unsigned)((P + kPixPerBlock - 1) / kPixPerBlock);
    hipStream_t st = (hipStream_t)stream;
    if (dxdir) {
        SMLVM_LR_DISPATCH_R(R, SMLVM_LR_DISPATCH_Y(y_kind,
            hipLaunchKernelGGL((lowrank_ce_fwd_kernel<RR, YY, true>), dim3(grid), dim3(256), 0, st, x, w, b, y,
                               lse, nll, dxdir, P, C)));
    } else {
        SMLVM_LR_DISPATCH_R(R, SMLVM_LR_DISPATCH_Y(y_kind,
            hipLaunchKernelGGL((lowrank_ce_fwd_kernel<RR, YY, false>), dim3(grid), dim3(256), 0, st, x, w, b, y,
                               lse, nll, (float*)nullptr, P, C)));
    }
    int rc2 = check_launch();
    if (rc2 != SMLVM_OK || !loss) return rc2;
    hipLaunchKernelGGL(row_sum_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, nll, loss, N, F);
    return check_launch();
}

extern "C" int smlvm_lowrank_ce_fwd(const float* x, const float* w, const float* b, const void* y,
                                    int32_t y_kind, float* lse, float* nll, float* loss, int64_t N,
                                    int32_t F, int32_t C, int32_t R, void* stream)
{
    return lowrank_ce_fwd_go(x, w, b, y, y_kind, lse, nll, loss, nullptr, N, F, C, R, stream);
}

extern "C" int smlvm_lowrank_ce_fwd_dxdir(const float* x, const float* w, const float* b, const void* y,
                                          int32_t y_kind, float* lse, float* nll, float* loss,
                                          float* dxdir, int64_t N, int32_t F, int32_t C, int32_t R,
                                          void* stream)
{
    if (N > 0 && !dxdir) return SMLVM_ERR_INVALID;
    return lowrank_ce_fwd_go(x, w, b, y, y_kind, lse, nll, loss, dxdir, N, F, C, R, stream);
}

extern "C" size_t smlvm_lowrank_ce_bwd_workspace_bytes(int64_t N, int32_t F, int32_t C, int32_t R)
{
    if (lr_check(N, F, C, R) != SMLVM_OK || N == 0) return 0;
    return sizeof(float) * (size_t)dw_blocks(N * F) * C * (R + 1);
}

extern "C" int smlvm_lowrank_ce_bwd(const float* x, const float* w, const float* b, const void* y,
                                    int32_t y_kind, const float* lse, const float* gloss, float* dx,
                                    float* dw, float* db, int64_t N, int32_t F, int32_t C,
                                    int32_t R, void* workspace, size_t workspace_bytes,
                                    void* stream)
{
    const int rc = lr_check(N, F, C, R);
    if (rc != SMLVM_OK) return rc;
    if (N == 0) return SMLVM_OK;  // NB: dw/db are left untouched for an empty batch
    if (!x || !w || !b || !y || !lse || !gloss) return SMLVM_ERR_INVALID;
    const int64_t P = N * F;
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        const unsigned grid = (unsigned)((P + kPixPerBlock - 1) / kPixPerBlock);
        SMLVM_LR_DISPATCH_R(R, SMLVM_LR_DISPATCH_Y(y_kind,
            hipLaunchKernelGGL((lowrank_ce_bwd_dx_kernel<RR, YY>), dim3(grid), dim3(256), 0, st, x, w,
                               b, y, lse, gloss, dx, P, F, C)));
        const int rc2 = check_launch();
        if (rc2 != SMLVM_OK) return rc2;
    }
    if (dw || db) {
        const int nblk = dw_blocks(P);
        if (!workspace || workspace_bytes < smlvm_lowrank_ce_bwd_workspace_bytes(N, F, C, R))
            return SMLVM_ERR_WORKSPACE;
        float* partial = static_cast<float*>(workspace);
        SMLVM_LR_DISPATCH_R(R, SMLVM_LR_DISPATCH_Y(y_kind,
            hipLaunchKernelGGL((lowrank_ce_bwd_dw_kernel<RR, YY>), dim3(nblk), dim3(256), 0, st, x, w,
                               b, y, lse, gloss, partial, P, F, C)));
        int rc2 = check_launch();
        if (rc2 != SMLVM_OK) return rc2;
        const int per = C * (R + 1);
        hipLaunchKernelGGL(lowrank_ce_reduce_kernel, dim3((per + 3) / 4), dim3(256), 0, st,
                           partial, nblk, C, R, dw, db);
        rc2 = check_launch();
        if (rc2 != SMLVM_OK) return rc2;
    }
    return SMLVM_OK;
}
