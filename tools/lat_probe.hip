// Latency probes for the SparseMAP solver's building blocks (one wave per CU, like the solver):
// dependent v_add_f64 chain, dependent v_fma_f64 chain, LDS broadcast read round trip.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lat_probe.hip -o tools/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k_add(double* out, double x, int n)
{
    double a = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { a += x; asm volatile("" : "+v"(a)); }
    }
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) out[64] = (double)(t1 - t0);
}
__global__ __launch_bounds__(64) void k_add2(double* out, double x, int n)
{
    double a = threadIdx.x, b = 1.0;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { a += x; b += x; asm volatile("" : "+v"(a), "+v"(b)); }
    }
    long long t1 = clock64();
    out[threadIdx.x] = a + b;
    if (threadIdx.x == 0) out[64] = (double)(t1 - t0);
}
__global__ __launch_bounds__(64) void k_addf(float* out, float x, int n)
{
    float a = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { a += x; asm volatile("" : "+v"(a)); }
    }
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0);
}
__global__ __launch_bounds__(64) void k_lds(double* out, int n)
{
    __shared__ int idx[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) idx[i] = (i * 7 + 3) & 1023;
    __syncthreads();
    int j = 0;
    long long t0 = clock64();
    for (int i = 0; i < n * 16; ++i) j = idx[j];     // dependent broadcast reads
    long long t1 = clock64();
    out[threadIdx.x] = j;
    if (threadIdx.x == 0) out[64] = (double)(t1 - t0);
}
int main()
{
    double* d; float* f;
    hipMalloc(&d, 80 * 8); hipMalloc(&f, 80 * 4);
    const int n = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch, bool isf) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double ticks;
        if (isf) { float t; hipMemcpy(&t, f + 64, 4, hipMemcpyDeviceToHost); ticks = t; }
        else hipMemcpy(&ticks, d + 64, 8, hipMemcpyDeviceToHost);
        printf("%-28s %8.3f ms  %10.0f ticks  -> %.1f ns/op, %.2f ticks/op, tick = %.2f ns\n", name, ms, ticks,
               ms * 1e6 / (n * 16.0), ticks / (n * 16.0), ms * 1e6 / ticks);
    };
    run("dependent v_add_f64", [&] { k_add<<<1, 64>>>(d, 1e-9, n); }, false);
    run("2 interleaved f64 chains", [&] { k_add2<<<1, 64>>>(d, 1e-9, n); }, false);
    run("dependent v_add_f32", [&] { k_addf<<<1, 64>>>(f, 1e-9f, n); }, true);
    run("dependent LDS read", [&] { k_lds<<<1, 64>>>(d, n); }, false);
    run("dependent v_add_f64 x256 CUs", [&] { k_add<<<256, 64>>>(d, 1e-9, n); }, false);
    return 0;
}
