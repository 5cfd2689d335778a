// Segmented log-sum-exp over the compacted support lists: replaces the
// per-sample Python loop of compute_importance_sampling
// (experiments/bit_vector-vae/train.py:325-330).  One 16-lane group per segment (supports are a
// few dozen entries), two passes over the segment (max, then sum of exp), HBM bound and tiny.
#include "common.hpp"

#include <math.h>

namespace smlvm {

__global__ __launch_bounds__(256) void segmented_lse_kernel(const float* __restrict__ v,
                                                            const int64_t* __restrict__ off,
                                                            float shift, float* __restrict__ out,
                                                            int64_t R)
{
    const int gl = threadIdx.x & 15;
    int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool live = r < R;
    if (!live) r = R - 1;   // keep whole DPP groups busy
    const int64_t lo = off[r], hi = off[r + 1];
    float mx = -INFINITY;
    for (int64_t j = lo + gl; j < hi; j += 16) mx = fmaxf(mx, v[j] + shift);
    mx = group_reduce<16>(mx, OpMaxF{});
    float s = 0.0f;
    for (int64_t j = lo + gl; j < hi; j += 16) s += expf(v[j] + shift - mx);
    s = group_reduce<16>(s, OpSum{});
    if (live && gl == 0) out[r] = (hi > lo) ? mx + logf(s) : -INFINITY;
}

}  // namespace smlvm

using namespace smlvm;

extern "C" int smlvm_segmented_logsumexp(const float* v, const int64_t* row_offsets, float shift,
                                         float* out, int64_t R, void* stream)
{
    if (R < 0) return SMLVM_ERR_INVALID;
    if (R == 0) return SMLVM_OK;
    if (!row_offsets || !out) return SMLVM_ERR_INVALID;   // v may be NULL when every segment is empty
    hipLaunchKernelGGL(segmented_lse_kernel, dim3((unsigned)((R * 16 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, v, row_offsets, shift, out, R);
    return check_launch();
}
