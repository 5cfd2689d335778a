// Cycle cost of the solver's building blocks in isolation (one wave, LDS resident operands).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w tools/sm_parts.hip -o tools/sm_parts
#include "../sparse-marginalization-lvm_amd/csrc/sparsemap.hip"
#include <cstdio>
namespace smlvm { thread_local int g_last_hip_error = 0; }
using namespace smlvm;

template <int LD, int NB>
__global__ __launch_bounds__(64) void parts(double* out, int m, int n, int reps)
{
    extern __shared__ __align__(16) char sm[];
    double* A = (double*)sm;                       // [roundb(LD,NB)][LD]
    double* v0 = A + ((LD + 7) & ~7) * LD;          // 4 vectors of 160
    double* v1 = v0 + 160; double* v2 = v1 + 160; double* v3 = v2 + 160;
    uint32_t* cfg = (uint32_t*)(v3 + 160);         // [144][4]
    const int lane = threadIdx.x;
    for (int i = lane; i < ((LD + 7) & ~7) * LD; i += 64) A[i] = 1e-3 * (i % 17);
    for (int i = lane; i < 160; i += 64) { v0[i] = i < m ? 0.01 * i : 0.0; v1[i] = i < n ? 0.5 : 0.0; v2[i] = i < n ? 0.25 : 0.0; v3[i] = 0.0; }
    for (int i = lane; i < 144 * 4; i += 64) cfg[i] = 0x9e3779b9u * (i + 1);
    __syncthreads();
    const int cnt = ((m + 8) & ~7), n8 = ((n + 7) & ~7), m8 = ((m + 7) & ~7);
    double sink = 0.0;
    long long t[8];
    t[0] = clock64();
    for (int r = 0; r < reps; ++r) { sink += matvec_col<LD>(A + (lane < LD ? lane : 0), v0, cnt); asm volatile("" : "+v"(sink)); }
    t[1] = clock64();
    for (int r = 0; r < reps; ++r) {
        const double* const arr[2] = {v1, v2}; double acc[2];
        serial_sums<2>(arr, n8, acc); sink += acc[0] + acc[1]; asm volatile("" : "+v"(sink));
    }
    t[2] = clock64();
    for (int r = 0; r < reps; ++r) {
        double mon[2], moff[2];
        marginals<4, 2>(cfg, v0, m8, lane, mon, moff); sink += mon[0] + moff[1] + mon[1] + moff[0]; asm volatile("" : "+v"(sink));
    }
    t[3] = clock64();
    for (int r = 0; r < reps; ++r) {   // rank-one update of the live block (lane owns a column)
        if (lane <= m) {
            double* col = A + lane;
            const double dj = 1e-9 * lane;
            for (int i0 = 0; i0 < cnt; i0 += 8) {
                double a[8], tt[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = col[(i0 + u) * LD];
                ld8(v0 + i0, tt);
#pragma unroll
                for (int u = 0; u < 8; ++u) { const double pr = tt[u] * dj; a[u] = a[u] + pr; }
#pragma unroll
                for (int u = 0; u < 8; ++u) col[(i0 + u) * LD] = a[u];
            }
        }
        __syncthreads();
    }
    t[4] = clock64();
    for (int r = 0; r < reps; ++r) { __syncthreads(); asm volatile("" : "+v"(sink)); }
    t[5] = clock64();
    if (lane == 0) { for (int k = 0; k < 5; ++k) out[k] = (double)(t[k + 1] - t[k]) / reps; out[5] = sink; }
}

// Register-resident variant for the first tier (m <= 63): lane c owns column c of the inverse in
// 64 fp64 registers, vectors are lane-distributed and broadcast with v_readlane.
__device__ __forceinline__ double rl(double v, int lane) { return readlane_f64(v, lane); }

__global__ __launch_bounds__(64) void parts_reg(double* out, int m, int reps)
{
    const int lane = threadIdx.x;
    double A[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) A[i] = 1e-3 * ((i * 64 + lane) % 17);
    double bvec = lane <= m ? 0.01 * lane : 0.0;
    const int cnt = (m + 8) & ~7;
    double sink = 0.0;
    long long t[4];
    t[0] = clock64();
    for (int r = 0; r < reps; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (b * 8 < cnt) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { const double tt = A[b * 8 + u] * rl(bvec, b * 8 + u); acc += tt; }
            }
        }
        sink += acc; asm volatile("" : "+v"(sink)); bvec += 1e-12 * sink;
    }
    t[1] = clock64();
    for (int r = 0; r < reps; ++r) {
        const double dj = 1e-9 * lane + 1e-15 * sink;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (b * 8 < cnt) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { const double pr = rl(bvec, b * 8 + u) * dj; A[b * 8 + u] = A[b * 8 + u] + pr; }
            }
        }
        asm volatile("" : "+v"(sink));
    }
    t[2] = clock64();
    for (int r = 0; r < reps; ++r) {   // serial sum over 128 lane-distributed values, 2 chains
        double a0 = 0.0, a1 = 0.0;
        const double v0 = bvec, v1 = bvec * 0.5, w0 = bvec + 1.0, w1 = bvec + 2.0;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) { a0 += rl(v0, k); a1 += rl(w0, k); }
#pragma unroll 8
        for (int k = 0; k < 64; ++k) { a0 += rl(v1, k); a1 += rl(w1, k); }
        sink += a0 + a1; asm volatile("" : "+v"(sink));
    }
    t[3] = clock64();
#pragma unroll
    for (int i = 0; i < 64; ++i) sink += A[i];
    if (lane == 0) { for (int k = 0; k < 3; ++k) out[k] = (double)(t[k + 1] - t[k]) / reps; }
    out[8 + lane] = sink;
}

static void go_reg(int m)
{
    double* d; hipMalloc(&d, 8 * 80);
    parts_reg<<<1, 64>>>(d, m, 200);
    double h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("registers m=%d: matvec %.0f  rank1-update %.0f  serial_sums<2>(128 via readlane) %.0f cycles (%s)\n", m, h[0], h[1], h[2],
           hipGetErrorString(hipGetLastError()));
}

template <int LD, int NB>
static void go(int m, int n)
{
    double* d; hipMalloc(&d, 64);
    size_t lds = (size_t)((LD + 7) & ~7) * LD * 8 + 4 * 160 * 8 + 144 * 16;
    hipFuncSetAttribute((const void*)parts<LD, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    parts<LD, NB><<<1, 64, lds>>>(d, m, n, 200);
    double h[6]; hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
    printf("LD=%d NB=%d m=%d n=%d: matvec %.0f  serial_sums<2>(n) %.0f  marginals %.0f  rank1-update %.0f  sync %.0f cycles  (%s)\n",
           LD, NB, m, n, h[0], h[1], h[2], h[3], h[4], hipGetErrorString(hipGetLastError()));
}
int main()
{
    go_reg(50); go_reg(30); go_reg(62);
    go<64, 8>(50, 128); go<64, 8>(30, 128); go<130, 8>(50, 128); go<130, 8>(100, 128); go<34, 8>(20, 32);
    return 0;
}
