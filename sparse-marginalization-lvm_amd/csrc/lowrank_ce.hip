// Next row after the hot path: the decoder side of the bit-vector VAE over the
// compacted support.  The reference's CategoricalGenerator (experiments/bit_vector-vae/archs.py:18-29)
// materialises logits [N, 784, 256] = X[N,784,R] . W[256,R]^T + b (R = out_rank = 5) and
// reconstruction_loss (experiments/bit_vector-vae/train.py:361-371) runs a 256-way cross entropy
// over them, summed over the 784 features: 0.8 MB of logits per support row, written and re-read
// several times.  These kernels never materialise the logits: a logit costs R FMAs to recompute,
// so forward and backward recompute them from X (20 B per feature) and keep only the per-feature
// log-sum-exp.  Compute bound (fp32 VALU FMA + v_exp_f32); HBM traffic is ~32 B per (row, feature).
//
//   fwd   lane = one (row, feature): max pass + exp pass over the C classes; W and b are read with
//         wave-uniform addresses (scalar loads); then a per-row sum.
//   bwd   dX: lane = one (row, feature), softmax from the saved lse, dX = g (softmax - onehot) W
//         dW, db: thread = one class, loops over (row, feature) tiles staged in LDS; per-block
//         partials, then a deterministic fp64 reduction over blocks (no float atomics).
#include "common.hpp"

#include <math.h>

namespace smlvm {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

template <int R>
__device__ __forceinline__ float logit(const float (&x)[R], const float* __restrict__ w,
                                       const float* __restrict__ b, int c)
{
    float l = b[c];
#pragma unroll
    for (int r = 0; r < R; ++r) l = fmaf(x[r], w[c * R + r], l);
    return l;
}

template <typename YT>
__device__ __forceinline__ int label_at(const void* y, int64_t i)
{
    return (int)static_cast<const YT*>(y)[i];
}

template <int R, typename YT>
__global__ __launch_bounds__(256) void lowrank_ce_fwd_kernel(
    const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ b,
    const void* __restrict__ y, float* __restrict__ lse, float* __restrict__ nll, int64_t P, int C)
{
    int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = p < P;
    if (!live) p = P - 1;
    float x[R];
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = X[p * R + r];
    const int yy = label_at<YT>(y, p);
    float mx = -INFINITY, ly = 0.0f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const float l = logit<R>(x, W, b, c);
        mx = fmaxf(mx, l);
        ly = (c == yy) ? l : ly;
    }
    const float m2 = mx * kLog2e;
    float s = 0.0f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const float l = logit<R>(x, W, b, c);
        s += exp2f(fmaf(l, kLog2e, -m2));
    }
    const float L = fmaf(log2f(s), kLn2, mx);
    if (live) {
        lse[p] = L;
        nll[p] = L - ly;
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

template <int R, typename YT>
__global__ __launch_bounds__(256) void lowrank_ce_bwd_dx_kernel(
    const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ b,
    const void* __restrict__ y, const float* __restrict__ lse, const float* __restrict__ gloss,
    float* __restrict__ dX, int64_t P, int F, int C)
{
    int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = p < P;
    if (!live) p = P - 1;
    float x[R], dx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { x[r] = X[p * R + r]; dx[r] = 0.0f; }
    const int yy = label_at<YT>(y, p);
    const float g = gloss[p / F];
    const float L2 = lse[p] * kLog2e;
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const float l = logit<R>(x, W, b, c);
        const float pr = exp2f(fmaf(l, kLog2e, -L2));
        const float dl = pr - ((c == yy) ? 1.0f : 0.0f);
#pragma unroll
        for (int r = 0; r < R; ++r) dx[r] = fmaf(dl, W[c * R + r], dx[r]);
    }
    if (live) {
#pragma unroll
        for (int r = 0; r < R; ++r) dX[p * R + r] = g * dx[r];
    }
}

constexpr int kDwTile = 64;       // (row, feature) pairs staged per step
constexpr int kDwMaxBlocks = 2048;

template <int R, typename YT>
__global__ __launch_bounds__(256) void lowrank_ce_bwd_dw_kernel(
    const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ b,
    const void* __restrict__ y, const float* __restrict__ lse, const float* __restrict__ gloss,
    float* __restrict__ partial, int64_t P, int F, int C)
{
    __shared__ float xs[kDwTile * R];
    __shared__ float ls[kDwTile];
    __shared__ float gs[kDwTile];
    __shared__ int ys[kDwTile];
    const int c = threadIdx.x;
    const bool mine = c < C;
    float w[R], acc[R + 1];
    float bc = 0.0f;
#pragma unroll
    for (int r = 0; r < R; ++r) { w[r] = mine ? W[c * R + r] : 0.0f; acc[r] = 0.0f; }
    acc[R] = 0.0f;
    if (mine) bc = b[c];
    const int64_t tiles = (P + kDwTile - 1) / kDwTile;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t p0 = t * kDwTile;
        const int cnt = (int)((P - p0 < kDwTile) ? (P - p0) : kDwTile);
        __syncthreads();  // previous tile fully consumed
        for (int i = threadIdx.x; i < cnt * R; i += 256) xs[i] = X[p0 * R + i];
        if (threadIdx.x < cnt) {
            const int64_t p = p0 + threadIdx.x;
            ls[threadIdx.x] = lse[p] * kLog2e;
            gs[threadIdx.x] = gloss[p / F];
            ys[threadIdx.x] = label_at<YT>(y, p);
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < cnt; ++i) {
            float l = bc;
#pragma unroll
            for (int r = 0; r < R; ++r) l = fmaf(xs[i * R + r], w[r], l);
            const float pr = exp2f(fmaf(l, kLog2e, -ls[i]));
            const float dl = gs[i] * (pr - ((c == ys[i]) ? 1.0f : 0.0f));
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = fmaf(dl, xs[i * R + r], acc[r]);
            acc[R] += dl;
        }
    }
    if (mine) {
        float* o = partial + ((int64_t)blockIdx.x * C + c) * (R + 1);
#pragma unroll
        for (int r = 0; r <= R; ++r) o[r] = acc[r];
    }
}

// dW[c, r] / db[c] = sum over blocks of the partials, in block order, fp64
__global__ __launch_bounds__(256) void lowrank_ce_reduce_kernel(const float* __restrict__ partial,
                                                                int nblk, int C, int R,
                                                                float* __restrict__ dW,
                                                                float* __restrict__ db)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int per = C * (R + 1);
    if (j >= per) return;
    double acc = 0.0;
    for (int k = 0; k < nblk; ++k) acc += (double)partial[(int64_t)k * per + j];
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

extern "C" int smlvm_lowrank_ce_fwd(const float* x, const float* w, const float* b, const void* y,
                                    int32_t y_kind, float* lse, float* nll, float* loss, int64_t N,
                                    int32_t F, int32_t C, int32_t R, void* stream)
{
    const int rc = lr_check(N, F, C, R);
    if (rc != SMLVM_OK) return rc;
    if (N == 0) return SMLVM_OK;
    if (!x || !w || !b || !y || !lse || !nll) return SMLVM_ERR_INVALID;
    const int64_t P = N * F;
    const unsigned grid = (unsigned)((P + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    SMLVM_LR_DISPATCH_R(R, SMLVM_LR_DISPATCH_Y(y_kind,
        hipLaunchKernelGGL((lowrank_ce_fwd_kernel<RR, YY>), dim3(grid), dim3(256), 0, st, x, w, b, y,
                           lse, nll, P, C)));
    int rc2 = check_launch();
    if (rc2 != SMLVM_OK || !loss) return rc2;
    hipLaunchKernelGGL(row_sum_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, nll, loss, N, F);
    return check_launch();
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
        const unsigned grid = (unsigned)((P + 255) / 256);
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
        hipLaunchKernelGGL(lowrank_ce_reduce_kernel, dim3((per + 255) / 256), dim3(256), 0, st,
                           partial, nblk, C, R, dw, db);
        rc2 = check_launch();
        if (rc2 != SMLVM_OK) return rc2;
    }
    return SMLVM_OK;
}
