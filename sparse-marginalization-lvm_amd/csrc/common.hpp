// Shared device/host helpers for the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sparsemarg.h"

namespace smlvm {

// ------------------------------------------------------------------------------------------ host
extern thread_local int g_last_hip_error;

inline int check_launch()
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return SMLVM_ERR_HIP;
    }
    return SMLVM_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------------------------------------------------------------------------------- device
// DPP controls (gfx9): quad_perm = 0x00..0xFF, row_mirror = 0x140, row_half_mirror = 0x141.
constexpr int DPP_QUAD_XOR1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;   // quad_perm [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_SHR1 = 0x111;

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, dpp_i32<CTRL>(__builtin_bit_cast(int, v)));
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    long long b = __builtin_bit_cast(long long, v);
    int lo = dpp_i32<CTRL>((int)(b & 0xffffffffll));
    int hi = dpp_i32<CTRL>((int)(b >> 32));
    b = ((long long)hi << 32) | (unsigned int)lo;
    return __builtin_bit_cast(double, b);
}

__device__ __forceinline__ float readlane_f32(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    b = ((long long)hi << 32) | (unsigned int)lo;
    return __builtin_bit_cast(double, b);
}

struct OpSum { template <class T> __device__ __forceinline__ T operator()(T a, T b) const { return a + b; } };
struct OpMaxF { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };
struct OpMinF { __device__ __forceinline__ float operator()(float a, float b) const { return fminf(a, b); } };
struct OpMaxI { __device__ __forceinline__ int operator()(int a, int b) const { return a > b ? a : b; } };

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS operations of one wave execute in order; this only pins the compiler's order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// All-reduce over groups of G consecutive lanes (G in {8,16,32,64}) of one wave: every lane of a
// group ends with the group's result, bitwise identical in all of them (each step combines the
// same two operands, commutatively; steps past one DPP row read whole-row results).
template <int G, class Op>
__device__ __forceinline__ float group_reduce(float v, Op op)
{
    v = op(v, dpp_f32<DPP_QUAD_XOR1>(v));
    v = op(v, dpp_f32<DPP_QUAD_XOR2>(v));
    v = op(v, dpp_f32<DPP_ROW_HALF_MIRROR>(v));
    if constexpr (G >= 16) v = op(v, dpp_f32<DPP_ROW_MIRROR>(v));
    if constexpr (G == 32) v = op(v, __shfl_xor(v, 16));
    if constexpr (G == 64) {
        float r0 = readlane_f32(v, 0), r1 = readlane_f32(v, 16), r2 = readlane_f32(v, 32),
              r3 = readlane_f32(v, 48);
        v = op(op(r0, r1), op(r2, r3));
    }
    return v;
}
template <int G, class Op>
__device__ __forceinline__ int group_reduce(int v, Op op)
{
    v = op(v, dpp_i32<DPP_QUAD_XOR1>(v));
    v = op(v, dpp_i32<DPP_QUAD_XOR2>(v));
    v = op(v, dpp_i32<DPP_ROW_HALF_MIRROR>(v));
    if constexpr (G >= 16) v = op(v, dpp_i32<DPP_ROW_MIRROR>(v));
    if constexpr (G == 32) v = op(v, __shfl_xor(v, 16));
    if constexpr (G == 64) {
        int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16),
            r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
        v = op(op(r0, r1), op(r2, r3));
    }
    return v;
}
template <int G, class Op>
__device__ __forceinline__ double group_reduce(double v, Op op)
{
    v = op(v, dpp_f64<DPP_QUAD_XOR1>(v));
    v = op(v, dpp_f64<DPP_QUAD_XOR2>(v));
    v = op(v, dpp_f64<DPP_ROW_HALF_MIRROR>(v));
    if constexpr (G >= 16) v = op(v, dpp_f64<DPP_ROW_MIRROR>(v));
    if constexpr (G == 32) v = op(v, __shfl_xor(v, 16));
    if constexpr (G == 64) {
        double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32),
               r3 = readlane_f64(v, 48);
        v = op(op(r0, r1), op(r2, r3));
    }
    return v;
}

}  // namespace smlvm
