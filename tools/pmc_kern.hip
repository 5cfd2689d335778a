// Launches ONE entry point of the library `reps` times on device-generated inputs, so that a
// rocprofv3 --pmc / --kernel-trace pass over this process sees only that kernel (plus the one
// forward launch that prepares p and the support lists for the other kernels).
//   pmc_kern <lib.so> <what> [R K sigma reps]
//   what: fwd | bwd | count | count_cols | offsets | compact | compact_cols | wlr |
//         sparsemap (R = samples, K = bits, sigma = score scale; budget via PMC_BUDGET)
// Prints "<what> <ms per call>" (HIP events).  Experiment / profiling driver only.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

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
// rows r % 4 == 0: all 0.25; 1: -1 with one 3.0; 2: -1 with one 0.0 (exact tie at the threshold); 3: left as is
__global__ void k_adversarial(float* x, int64_t R, int K)
{
    const size_t n = (size_t)R * K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int64_t r = (int64_t)(i / K);
        const int c = (int)(i - (size_t)r * K);
        const int kind = (int)(r & 3);
        if (kind == 0) x[i] = 0.25f;
        else if (kind == 1) x[i] = (c == 7 % K) ? 3.0f : -1.0f;
        else if (kind == 2) x[i] = (c == 0) ? 0.0f : -1.0f;
    }
}
__global__ void k_unif64(double* x, size_t n, unsigned long long seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        x[i] = (mix(i + seed) >> 5) * (1.0 / 134217728.0);
}

template <class F> F sym(void* h, const char* name)
{
    F f = (F)dlsym(h, name);
    if (!f) { fprintf(stderr, "missing symbol %s\n", name); exit(1); }
    return f;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: pmc_kern lib.so what [R K sigma reps]\n"); return 2; }
    const char* what = argv[2];
    const int64_t R = argc > 3 ? atoll(argv[3]) : 1048576;
    const int K = argc > 4 ? atoi(argv[4]) : 4096;
    const float sigma = argc > 5 ? atof(argv[5]) : 1.0f;
    const int reps = argc > 6 ? atoi(argv[6]) : 3;
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](auto&& f) {
        // timing runs (>= 10 calls) warm up first; counter passes (3 calls) must not add dispatches
        for (int i = 0; i < (reps >= 10 ? 3 : 0); ++i) f();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) { int rc = f(); if (rc) { printf("%s rc=%d\n", what, rc); exit(1); } }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s %.4f ms per call (%d calls, R=%lld K=%d sigma=%g)\n", what, ms / reps, reps, (long long)R, K, sigma);
    };

    if (!strcmp(what, "sparsemap")) {
        typedef size_t (*ws_t)(int64_t, int32_t, int32_t);
        typedef int (*fwd_t)(const float*, const double*, int32_t, int32_t, int64_t, int32_t, uint32_t*, float*,
                             int32_t*, double*, int32_t*, int32_t*, void*, size_t, void*);
        auto wsb = sym<ws_t>(h, "smlvm_sparsemap_fwd_workspace_bytes");
        auto fwd = sym<fwd_t>(h, "smlvm_sparsemap_fwd");
        const int n = K, cap = n + 1, W = (n + 31) / 32, ld = n + 2;
        const int budget = getenv("PMC_BUDGET") ? atoi(getenv("PMC_BUDGET")) : 0;
        float* sc; double* init; uint32_t* bits; float* p; int32_t *alen, *iters, *status; double* inva; void* ws;
        CK(hipMalloc(&sc, R * n * 4)); CK(hipMalloc(&init, R * 2 * n * 8)); CK(hipMalloc(&bits, R * cap * W * 4));
        CK(hipMalloc(&p, R * cap * 4)); CK(hipMalloc(&alen, R * 4)); CK(hipMalloc(&iters, R * 4)); CK(hipMalloc(&status, R * 4));
        CK(hipMalloc(&inva, (size_t)R * ld * ld * 8));
        const size_t wbytes = wsb(R, n, 1);
        CK(hipMalloc(&ws, wbytes ? wbytes : 8));
        hipLaunchKernelGGL(k_gauss, dim3(1024), dim3(256), 0, 0, sc, (size_t)R * n, sigma, 1ull);
        hipLaunchKernelGGL(k_unif64, dim3(1024), dim3(256), 0, 0, init, (size_t)R * 2 * n, 9ull);
        CK(hipDeviceSynchronize());
        timed([&] { return fwd(sc, init, budget, 300, R, n, bits, p, alen, inva, iters, status, ws, wbytes, nullptr); });
        int32_t* hi = (int32_t*)malloc(R * 4);
        CK(hipMemcpy(hi, iters, R * 4, hipMemcpyDeviceToHost));
        long long tot = 0; for (int64_t i = 0; i < R; ++i) tot += hi[i];
        printf("active-set iterations per call: %lld\n", tot);
        return 0;
    }

    typedef int (*fwd_cols_t)(const float*, float*, int32_t*, int32_t*, float*, float*, int32_t*, int, uint32_t*, int64_t, int, void*);
    typedef int (*bwd_cols_t)(const float*, const int32_t*, int, const uint32_t*, const float*, const int32_t*, float*, int64_t, int, void*);
    typedef int (*count_t)(const float*, int32_t*, int64_t, int32_t, void*);
    typedef int (*count_cols_t)(const float*, const int32_t*, int32_t, int32_t*, int64_t, int32_t, void*);
    typedef size_t (*szr_t)(int64_t);
    typedef size_t (*mask_t)(int64_t, int32_t);
    typedef int (*offsets_t)(const int32_t*, int64_t*, int64_t, void*, size_t, void*);
    typedef int (*compact_t)(const float*, const int64_t*, int32_t*, int32_t*, float*, int64_t, int32_t, void*);
    typedef int (*compact_cols_t)(const float*, const int32_t*, int32_t, const int64_t*, int32_t*, int32_t*, float*, int64_t, int32_t, void*);
    typedef int (*wlr_t)(const float*, const float*, const int64_t*, float*, float*, float, int64_t, int64_t, void*, size_t, void*);
    auto fwdc = sym<fwd_cols_t>(h, "smlvm_sparsemax_fwd_cols");
    auto bwdc = sym<bwd_cols_t>(h, "smlvm_sparsemax_bwd_cols");
    const int cap = K > 4096 ? 128 : 32;   // as lvmhelpers._native.support_cols_cap
    const size_t n = (size_t)R * K;
    float *x, *p; int32_t *supp, *nnz, *cols; uint32_t* mask = nullptr;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&supp, R * 4)); CK(hipMalloc(&nnz, R * 4));
    CK(hipMalloc(&cols, (size_t)R * cap * 4));
    const size_t mb = sym<mask_t>(h, "smlvm_sparsemax_support_mask_bytes")(R, K);
    if (mb) { CK(hipMalloc(&mask, mb)); CK(hipMemset(mask, 0, mb)); }
    hipLaunchKernelGGL(k_gauss, dim3(4096), dim3(256), 0, 0, x, n, sigma, 1ull);
    if (!strcmp(what, "fwdadv"))   // SURVEY.md 8(d) S1 adversarial rows, a quarter each (bench.py's mix)
        hipLaunchKernelGGL(k_adversarial, dim3(4096), dim3(256), 0, 0, x, R, K);
    CK(hipDeviceSynchronize());
    if (!strcmp(what, "fwdadv")) {
        timed([&] { return fwdc(x, p, supp, nullptr, nullptr, nullptr, cols, cap, mask, R, K, nullptr); });
        return 0;
    }
    if (!strcmp(what, "efwd")) {   // entmax15 forward
        typedef int (*e15_t)(const float*, float*, int32_t*, float*, int64_t, int32_t, void*);
        auto f = sym<e15_t>(h, "smlvm_entmax15_fwd");
        timed([&] { return f(x, p, nullptr, nullptr, R, K, nullptr); });
        return 0;
    }
    if (!strcmp(what, "fused")) {  // single-pass compaction from the forward's column + value lists
        typedef int (*fwdl_t)(const float*, float*, int32_t*, int32_t*, float*, float*, int32_t*, float*, int, uint32_t*, int64_t, int, void*);
        typedef int (*fused_t)(const float*, const int32_t*, const float*, int32_t, int64_t*, int32_t*, int32_t*, float*, int64_t,
                               int64_t, int32_t, void*, size_t, void*);
        auto fl = sym<fwdl_t>(h, "smlvm_sparsemax_fwd_lists");
        auto ff = sym<fused_t>(h, "smlvm_support_compact_fused");
        auto wsz = sym<szr_t>(h, "smlvm_support_compact_fused_workspace_bytes");
        float* vals_l; CK(hipMalloc(&vals_l, (size_t)R * cap * 4));
        if (fl(x, p, supp, nnz, nullptr, nullptr, cols, vals_l, cap, mask, R, K, nullptr)) { printf("setup fwd_lists failed\n"); return 1; }
        const int64_t n_cap = R * (int64_t)cap;
        int64_t* off; void* ws; const size_t wb = wsz(R);
        int32_t *rows, *cc; float* vals;
        CK(hipMalloc(&off, (R + 1) * 8)); CK(hipMalloc(&ws, wb));
        CK(hipMalloc(&rows, n_cap * 4)); CK(hipMalloc(&cc, n_cap * 4)); CK(hipMalloc(&vals, n_cap * 4));
        CK(hipDeviceSynchronize());
        timed([&] { return ff(p, cols, vals_l, cap, off, rows, cc, vals, n_cap, R, K, ws, wb, nullptr); });
        int64_t N; CK(hipMemcpy(&N, off + R, 8, hipMemcpyDeviceToHost));
        printf("N = %lld\n", (long long)N);
        return 0;
    }
    // one forward prepares p, supp, nnz and the support lists for everything else
    // (for the forward target the same instantiation as the timed calls: the variant with optional
    // outputs spills to scratch at K >= 4096, which would be averaged into the forward's traffic)
    if (fwdc(x, p, supp, strcmp(what, "fwd") ? nnz : nullptr, nullptr, nullptr, cols, cap, mask, R, K, nullptr)) { printf("setup fwd failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    if (!strcmp(what, "fwd")) {
        timed([&] { return fwdc(x, p, supp, nullptr, nullptr, nullptr, cols, cap, mask, R, K, nullptr); });
    } else if (!strcmp(what, "bwd")) {
        float *dy, *dx; CK(hipMalloc(&dy, n * 4)); CK(hipMalloc(&dx, n * 4));
        hipLaunchKernelGGL(k_gauss, dim3(4096), dim3(256), 0, 0, dy, n, 1.0f, 77ull);
        CK(hipDeviceSynchronize());
        timed([&] { return bwdc(dy, cols, cap, mask, p, supp, dx, R, K, nullptr); });
    } else if (!strcmp(what, "count")) {
        auto f = sym<count_t>(h, "smlvm_support_count");
        timed([&] { return f(p, nnz, R, K, nullptr); });
    } else if (!strcmp(what, "count_cols")) {
        auto f = sym<count_cols_t>(h, "smlvm_support_count_cols");
        timed([&] { return f(p, cols, cap, nnz, R, K, nullptr); });
    } else {
        auto wsz = sym<szr_t>(h, "smlvm_support_offsets_workspace_bytes");
        auto offs = sym<offsets_t>(h, "smlvm_support_offsets");
        int64_t* off; void* ws; const size_t wb = wsz(R);
        CK(hipMalloc(&off, (R + 1) * 8)); CK(hipMalloc(&ws, wb));
        if (!strcmp(what, "offsets")) {
            timed([&] { return offs(nnz, off, R, ws, wb, nullptr); });
            return 0;
        }
        if (offs(nnz, off, R, ws, wb, nullptr)) return 1;
        int64_t N; CK(hipMemcpy(&N, off + R, 8, hipMemcpyDeviceToHost));
        printf("N = %lld\n", (long long)N);
        int32_t *rows, *cc; float* vals;
        CK(hipMalloc(&rows, N * 4 + 4)); CK(hipMalloc(&cc, N * 4 + 4)); CK(hipMalloc(&vals, N * 4 + 4));
        if (!strcmp(what, "compact")) {
            auto f = sym<compact_t>(h, "smlvm_support_compact");
            timed([&] { return f(p, off, rows, cc, vals, R, K, nullptr); });
        } else if (!strcmp(what, "compact_cols")) {
            auto f = sym<compact_cols_t>(h, "smlvm_support_compact_cols");
            timed([&] { return f(p, cols, cap, off, rows, cc, vals, R, K, nullptr); });
        } else if (!strcmp(what, "wlr")) {
            auto f0 = sym<compact_t>(h, "smlvm_support_compact");
            if (f0(p, off, rows, cc, vals, R, K, nullptr)) return 1;
            float *comps, *total; CK(hipMalloc(&comps, N * 4 + 4)); CK(hipMalloc(&total, 4));
            hipLaunchKernelGGL(k_gauss, dim3(1024), dim3(256), 0, 0, comps, (size_t)N, 100.0f, 5ull);
            auto wr = sym<szr_t>(h, "smlvm_weighted_loss_reduce_workspace_bytes");
            void* w2; const size_t w2b = wr(N); CK(hipMalloc(&w2, w2b));
            auto f = sym<wlr_t>(h, "smlvm_weighted_loss_reduce_fwd");
            CK(hipDeviceSynchronize());
            timed([&] { return f(vals, comps, nullptr, nullptr, total, 1.0f / R, 0, N, w2, w2b, nullptr); });
        } else {
            fprintf(stderr, "unknown what %s\n", what);
            return 2;
        }
    }
    return 0;
}
