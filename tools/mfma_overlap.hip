// Does v_mfma_f32_32x32x2_f32 run beside VALU work of other waves (or of its own wave) on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_overlap tools/mfma_overlap.hip
// Modes: 0 = MFMA only, 1 = v_exp_f32 only, 2 = both in one wave, 3 = v_fma_f32 only, 4 = MFMA + fma.
// Grid: 256 CUs x 4 SIMDs x W waves per SIMD; per iteration 4 MFMAs (256 issue cycles) and/or 16 VALU ops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float e[16];
    for (int i = 0; i < 16; ++i) e[i] = seed * (threadIdx.x + i);
    const float wa = seed + threadIdx.x, xb = seed - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2 || MODE == 4) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, xb, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, xb, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, xb, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, xb, a3, 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
        }
        if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) e[i] = __builtin_fmaf(e[i], seed, 0.5f);
        }
    }
    float s = 0.0f;
    for (int i = 0; i < 16; ++i) s += e[i] + a0[i] + a1[i] + a2[i] + a3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int waves_per_simd, float* out)
{
    const int iters = 2000, blocks = 256 * waves_per_simd;   // one 256-thread block = 1 wave on each SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.001f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.001f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-12s waves/SIMD %d: %.3f ms = %.0f cycles per iteration (at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / iters);
}

int main()
{
    float* out; hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("mfma", w, out); run<1>("exp", w, out); run<2>("mfma+exp", w, out);
        run<3>("fma x64", w, out); run<4>("mfma+fma", w, out);
    }
    hipFree(out);
    return 0;
}
