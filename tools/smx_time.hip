// Times the sparsemax forward/backward entry points of a given build of the library on a
// device-generated [R,K] matrix: smx_time <lib.so> [R K sigma reps].  Experiment driver only.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline unsigned mix(unsigned long long i)
{
    i ^= i >> 33; i *= 0xff51afd7ed558ccdull; i ^= i >> 33; i *= 0xc4ceb9fe1a85ec53ull; i ^= i >> 33;
    return (unsigned)i;
}
__global__ void k_gauss(float* x, size_t n, float sigma, unsigned long long seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float u1 = (mix(2 * i + seed) + 1.0f) * (1.0f / 4294967808.0f);
        const float u2 = mix(2 * i + 1 + seed) * (1.0f / 4294967296.0f);
        x[i] = sigma * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
    }
}

typedef int (*fwd_cols_t)(const float*, float*, int32_t*, int32_t*, float*, float*, int32_t*, int, uint32_t*, int64_t, int, void*);
typedef int (*bwd_cols_t)(const float*, const int32_t*, int, const uint32_t*, const float*, const int32_t*, float*, int64_t, int, void*);
typedef int (*fwd_t)(const float*, float*, int32_t*, int32_t*, float*, float*, int64_t, int, void*);
typedef int (*bwd_t)(const float*, const float*, const int32_t*, float*, int64_t, int, void*);

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: smx_time lib.so [R K sigma reps cap]\n"); return 2; }
    const int64_t R = argc > 2 ? atoll(argv[2]) : 1048576;
    const int K = argc > 3 ? atoi(argv[3]) : 4096;
    const float sigma = argc > 4 ? atof(argv[4]) : 1.0f;
    const int reps = argc > 5 ? atoi(argv[5]) : 10;
    const int cap = argc > 6 ? atoi(argv[6]) : 32;
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
    fwd_cols_t fwdc = (fwd_cols_t)dlsym(h, "smlvm_sparsemax_fwd_cols");
    bwd_cols_t bwdc = (bwd_cols_t)dlsym(h, "smlvm_sparsemax_bwd_cols");
    fwd_t fwd = (fwd_t)dlsym(h, "smlvm_sparsemax_fwd");
    bwd_t bwd = (bwd_t)dlsym(h, "smlvm_sparsemax_bwd");
    if (!fwd || !bwd) { fprintf(stderr, "missing symbols\n"); return 1; }
    const size_t n = (size_t)R * K;
    float *x, *p, *dy, *dx; int32_t *supp, *cols;
    // SMX_PAD=<bytes>: shift p (and dx) inside their allocations, to probe how the relative placement
    // of the input and output streams affects the rate
    const size_t pad = getenv("SMX_PAD") ? (size_t)atoll(getenv("SMX_PAD")) : 0;
    if (getenv("SMX_ARENA")) {   // one allocation, the four tensors back to back (+ pad between them)
        char* arena;
        CK(hipMalloc(&arena, 4 * (n * 4 + pad)));
        x = (float*)arena; p = (float*)(arena + (n * 4 + pad)); dy = (float*)(arena + 2 * (n * 4 + pad));
        dx = (float*)(arena + 3 * (n * 4 + pad));
    } else {
        if (getenv("SMX_ORDER")) {   // the order bench.py allocates in: x, dy, then p, dx
            CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&dy, n * 4)); CK(hipMalloc(&p, n * 4 + pad)); CK(hipMalloc(&dx, n * 4 + pad));
        } else {
            CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&p, n * 4 + pad)); CK(hipMalloc(&dy, n * 4)); CK(hipMalloc(&dx, n * 4 + pad));
        }
        p = (float*)((char*)p + pad); dx = (float*)((char*)dx + pad);
    }
    printf("x %p p %p dy %p dx %p\n", (void*)x, (void*)p, (void*)dy, (void*)dx);
    CK(hipMalloc(&supp, R * 4)); CK(hipMalloc(&cols, (size_t)R * cap * 4));
    uint32_t* mask = nullptr;
    if (K > 1024 && K <= 4096 && !getenv("SMX_NO_MASK")) CK(hipMalloc(&mask, (size_t)R * 512));
    hipLaunchKernelGGL(k_gauss, dim3(4096), dim3(256), 0, 0, x, n, sigma, 1ull);
    hipLaunchKernelGGL(k_gauss, dim3(4096), dim3(256), 0, 0, dy, n, 1.0f, 77ull);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](const char* name, auto&& f) {
        for (int i = 0; i < 3; ++i) { int rc = f(); if (rc) { printf("%s rc=%d\n", name, rc); return; } }
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-10s %.4f ms\n", name, ms / reps);
    };
    time_it("fwd", [&] { return fwd(x, p, supp, nullptr, nullptr, nullptr, R, K, nullptr); });
    time_it("bwd", [&] { return bwd(dy, p, supp, dx, R, K, nullptr); });
    if (fwdc && bwdc) {
        time_it("fwd_cols", [&] { return fwdc(x, p, supp, nullptr, nullptr, nullptr, cols, cap, mask, R, K, nullptr); });
        time_it("bwd_cols", [&] { return bwdc(dy, cols, cap, mask, p, supp, dx, R, K, nullptr); });
    }
    return 0;
}
