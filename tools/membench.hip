// HBM ceiling probes for the access shapes of the sparsemax kernels (float4 per lane, streaming).
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ o, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) o[i] = a[i];
}
// each block handles a contiguous tile of U float4 per thread (like one row per wave)
template <int U>
__global__ __launch_bounds__(256) void k_copy_tile(const float4* __restrict__ a, float4* __restrict__ o, size_t n)
{
    size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    const float4* src = a + wave * (64 * U);
    float4* dst = o + wave * (64 * U);
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[u * 64 + lane];
#pragma unroll
    for (int u = 0; u < U; ++u) dst[u * 64 + lane] = v[u];
}
template <int U>
__global__ __launch_bounds__(256) void k_2r1w_tile(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ o, size_t n)
{
    size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    const float4* s1 = a + wave * (64 * U);
    const float4* s2 = b + wave * (64 * U);
    float4* dst = o + wave * (64 * U);
    float4 v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { v[u] = s1[u * 64 + lane]; w[u] = s2[u * 64 + lane]; }
#pragma unroll
    for (int u = 0; u < U; ++u) dst[u * 64 + lane] = make_float4(v[u].x + w[u].x, v[u].y + w[u].y, v[u].z + w[u].z, v[u].w + w[u].w);
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, float* __restrict__ o, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t stride = (size_t)gridDim.x * 256;
    float s = 0;
    for (; i < n; i += stride) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) o[0] = s;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ o, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) o[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <int U, bool NTL, bool NTS, int BS>
__global__ __launch_bounds__(BS) void k_copy_tile_nt(const float4* __restrict__ a, float4* __restrict__ o, size_t n)
{
    size_t wave = ((size_t)blockIdx.x * BS + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    const float4* src = a + wave * (64 * U);
    float4* dst = o + wave * (64 * U);
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* s4 = reinterpret_cast<const f4*>(src);
    f4* d4 = reinterpret_cast<f4*>(dst);
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (NTL) v[u] = __builtin_nontemporal_load(s4 + u * 64 + lane); else v[u] = s4[u * 64 + lane];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (NTS) __builtin_nontemporal_store(v[u], d4 + u * 64 + lane); else d4[u * 64 + lane] = v[u];
    }
}

// block tile: the BS threads of a block jointly own BS*U consecutive float4; instruction u covers
// BS*16 contiguous bytes
template <int U, int BS, bool NT>
__global__ __launch_bounds__(BS) void k_copy_btile(const float4* __restrict__ a, float4* __restrict__ o, size_t n)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* s4 = reinterpret_cast<const f4*>(a) + (size_t)blockIdx.x * (BS * U);
    f4* d4 = reinterpret_cast<f4*>(o) + (size_t)blockIdx.x * (BS * U);
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (NT) v[u] = __builtin_nontemporal_load(s4 + u * BS + threadIdx.x); else v[u] = s4[u * BS + threadIdx.x];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (NT) __builtin_nontemporal_store(v[u], d4 + u * BS + threadIdx.x); else d4[u * BS + threadIdx.x] = v[u];
    }
}
template <int U, int BS, bool NT>
__global__ __launch_bounds__(BS) void k_2r1w_btile(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ o, size_t n)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* s4 = reinterpret_cast<const f4*>(a) + (size_t)blockIdx.x * (BS * U);
    const f4* t4 = reinterpret_cast<const f4*>(b) + (size_t)blockIdx.x * (BS * U);
    f4* d4 = reinterpret_cast<f4*>(o) + (size_t)blockIdx.x * (BS * U);
    f4 v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (NT) { v[u] = __builtin_nontemporal_load(s4 + u * BS + threadIdx.x); w[u] = __builtin_nontemporal_load(t4 + u * BS + threadIdx.x); }
        else { v[u] = s4[u * BS + threadIdx.x]; w[u] = t4[u * BS + threadIdx.x]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (NT) __builtin_nontemporal_store(v[u] + w[u], d4 + u * BS + threadIdx.x); else d4[u * BS + threadIdx.x] = v[u] + w[u];
    }
}

// XCD-aware: workgroup b runs on XCD b % 8; give each XCD one contiguous eighth of the tensor
template <int U, bool NT, int BS, int CH>
__global__ __launch_bounds__(BS) void k_copy_tile_xcd(const float4* __restrict__ a, float4* __restrict__ o, size_t n)
{
    const unsigned nb = gridDim.x, b = blockIdx.x;
    unsigned lb;
    if (CH == 0) lb = (b % 8) * (nb / 8) + b / 8;                     // whole eighths
    else { const unsigned g = b / (8 * CH), r = b % (8 * CH); lb = g * 8 * CH + (r % 8) * CH + r / 8; }  // chunks of CH blocks
    size_t wave = ((size_t)lb * BS + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* s4 = reinterpret_cast<const f4*>(a + wave * (64 * U));
    f4* d4 = reinterpret_cast<f4*>(o + wave * (64 * U));
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s4 + u * 64 + lane) : s4[u * 64 + lane];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d4 + u * 64 + lane); else d4[u * 64 + lane] = v[u]; }
}

// write-only tile: each wave zero-fills U KB (the sparse backward's shape)
template <int U, bool NT, int BS, bool REMAP>
__global__ __launch_bounds__(BS) void k_fill_tile(float4* __restrict__ o, size_t n)
{
    const unsigned nb = gridDim.x, b = blockIdx.x, per = nb >> 3;
    const unsigned lb = (REMAP && b < (per << 3)) ? (b & 7u) * per + (b >> 3) : b;
    size_t wave = ((size_t)lb * BS + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4* d4 = reinterpret_cast<f4*>(o + wave * (64 * U));
    const f4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(z, d4 + u * 64 + lane); else d4[u * 64 + lane] = z; }
}

template <class F>
static double timeit(F f, int reps = 10)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1e-3;
}

// persistent waves, software pipelined over 16 U-byte tiles: the loads of the next tile are issued
// chunk by chunk between the stores of the current one (MODE 0), or all before them (MODE 1)
template <int U, int MODE, bool REMAP>
__global__ __launch_bounds__(256) void k_copy_pipe(const float4* __restrict__ a, float4* __restrict__ o, size_t ntiles)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const size_t W = (size_t)gridDim.x * 4;
    size_t b = blockIdx.x;
    if (REMAP) { const size_t per = gridDim.x / 8; b = (blockIdx.x % 8) * per + blockIdx.x / 8; }
    size_t t = b * 4 + (threadIdx.x >> 6);
    const f4* s4 = reinterpret_cast<const f4*>(a);
    f4* d4 = reinterpret_cast<f4*>(o);
    f4 cur[U], nxt[U];
    if (t >= ntiles) return;
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = __builtin_nontemporal_load(s4 + t * (64 * U) + u * 64 + lane);
    for (;;) {
        const size_t tn = t + W;
        const bool more = tn < ntiles;
        if (MODE == 1 && more) {
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = __builtin_nontemporal_load(s4 + tn * (64 * U) + u * 64 + lane);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 0 && more) nxt[u] = __builtin_nontemporal_load(s4 + tn * (64 * U) + u * 64 + lane);
            __builtin_nontemporal_store(cur[u], d4 + t * (64 * U) + u * 64 + lane);
        }
        if (!more) break;
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        t = tn;
    }
}

// The K = 65536 forward's shape: 1024 threads hold a 256 KB tile in registers (16 float4 each),
// `work` rounds of dependent VALU work stand in for the threshold search, then the tile is stored.
// STAGE: persistent workgroups; while a tile is worked on and stored, the first half of the next
// tile is already on its way into LDS (global_load_lds_dwordx4, no registers involved).
__device__ __forceinline__ float burn(float v, int work)
{
    for (int i = 0; i < work; ++i) v = __builtin_fmaf(v, 1.0000001f, 1e-9f);
    return v;
}
template <bool STAGE>
__global__ __launch_bounds__(1024) void k_copy_wg(const float4* __restrict__ a, float4* __restrict__ o,
                                                   size_t ntiles, int work)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __align__(16) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const f4* s4 = reinterpret_cast<const f4*>(a);
    f4* d4 = reinterpret_cast<f4*>(o);
    f4 v[16];
    if (!STAGE) {
        const size_t t = blockIdx.x;
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = __builtin_nontemporal_load(s4 + t * 16384 + u * 1024 + threadIdx.x);
        v[0].x = burn(v[0].x + v[15].w, work);
#pragma unroll
        for (int u = 0; u < 16; ++u) __builtin_nontemporal_store(v[u], d4 + t * 16384 + u * 1024 + threadIdx.x);
        return;
    }
    size_t t = blockIdx.x;
    if (t >= ntiles) return;
    auto prefetch = [&](size_t tile) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            __builtin_amdgcn_global_load_lds(s4 + tile * 16384 + u * 1024 + wave * 64 + lane,
                                             (__attribute__((address_space(3))) void*)(lds + (u * 16 + wave) * 1024), 16, 0, 0);
    };
    prefetch(t);
    for (;;) {
#pragma unroll
        for (int u = 8; u < 16; ++u) v[u] = __builtin_nontemporal_load(s4 + t * 16384 + u * 1024 + threadIdx.x);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // the eight LDS loads were issued first
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f4*>(lds + (u * 16 + wave) * 1024 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const size_t tn = t + gridDim.x;
        if (tn < ntiles) prefetch(tn);
        v[0].x = burn(v[0].x + v[15].w, work);
#pragma unroll
        for (int u = 0; u < 16; ++u) __builtin_nontemporal_store(v[u], d4 + t * 16384 + u * 1024 + threadIdx.x);
        if (tn >= ntiles) break;
        t = tn;
    }
}

// Two passes over a tile of TILE_KB per workgroup of BS threads, several workgroups per CU (round 3
// probe for rows too wide for one wave): pass 1 streams the tile with ordinary loads (maximum of the
// tile, `work` rounds of dependent VALU work in one wave stand in for the threshold search), pass 2
// reads it again -- from the L2 / the memory-side cache if it is still there -- and stores it.
template <int BS>
__global__ __launch_bounds__(BS) void k_two_pass(const float4* __restrict__ a, float4* __restrict__ o,
                                                  size_t ntiles, int f4_per_tile, int work)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ float red[BS / 64];
    const f4* s4 = reinterpret_cast<const f4*>(a);
    f4* d4 = reinterpret_cast<f4*>(o);
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const f4* src = s4 + t * f4_per_tile;
        float m = -1e30f;
        for (int i = threadIdx.x; i < f4_per_tile; i += BS * 8) {
            f4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (i + u * BS < f4_per_tile) ? src[i + u * BS] : f4{0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 8; ++u) m = fmaxf(fmaxf(fmaxf(m, v[u].x), fmaxf(v[u].y, v[u].z)), v[u].w);
        }
        for (int d = 32; d; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x < 64) {
            float r = threadIdx.x < BS / 64 ? red[threadIdx.x] : -1e30f;
            for (int d = 32; d; d >>= 1) r = fmaxf(r, __shfl_xor(r, d));
            r = burn(r, work);
            if (threadIdx.x == 0) red[0] = r;
        }
        __syncthreads();
        const float tau = red[0];
        f4* dst = d4 + t * f4_per_tile;
        for (int i = threadIdx.x; i < f4_per_tile; i += BS * 8) {
            f4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (i + u * BS < f4_per_tile) ? __builtin_nontemporal_load(src + i + u * BS) : f4{0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i + u * BS < f4_per_tile) __builtin_nontemporal_store(v[u] - tau, dst + i + u * BS);
        }
        __syncthreads();
    }
}

int main(int argc, char** argv)
{
    size_t bytes = (size_t)16 << 30;  // per tensor
    if (argc > 1) bytes = (size_t)atof(argv[1]) * (1ull << 30);
    size_t n = bytes / 16;
    float4 *a, *b, *o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 1, bytes));
    double t;
    if (argc > 6) {
        for (int kb : {256, 64, 32}) {
            const int f4t = kb * 64;
            const size_t nt = n / f4t;
            for (int work : {0, 300}) {
#define TP(BS, G) t = timeit([&] { k_two_pass<BS><<<G, BS>>>(a, o, nt, f4t, work); }); \
    printf("two-pass %d KB tiles work=%d BS=%d grid=%d: %.2f TB/s (on 2 x bytes)\n", kb, work, BS, G, 2.0 * bytes / t / 1e12);
                TP(1024, 256) TP(1024, 512) TP(512, 512) TP(512, 1024) TP(256, 1024) TP(256, 2048)
            }
        }
        return 0;
    }
    if (argc > 5) {
        const size_t nt = n / 16384;
        CK(hipFuncSetAttribute((const void*)k_copy_wg<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        CK(hipFuncSetAttribute((const void*)k_copy_wg<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        for (int work : {0, 300, 1000}) {
            t = timeit([&] { k_copy_wg<false><<<(unsigned)nt, 1024>>>(a, o, nt, work); });
            printf("wg copy 256 KB tiles work=%d one tile per workgroup (two workgroups per CU): %.2f TB/s\n", work, 2.0 * bytes / t / 1e12);
            t = timeit([&] { k_copy_wg<false><<<(unsigned)nt, 1024, 128 * 1024>>>(a, o, nt, work); });
            printf("wg copy 256 KB tiles work=%d one tile per workgroup, one workgroup per CU (as the forward): %.2f TB/s\n", work, 2.0 * bytes / t / 1e12);
            for (int g : {256, 512}) {
                t = timeit([&] { k_copy_wg<true><<<g, 1024, 128 * 1024>>>(a, o, nt, work); });
                printf("wg copy 256 KB tiles work=%d staged through LDS, grid=%d: %.2f TB/s\n", work, g, 2.0 * bytes / t / 1e12);
            }
        }
        return 0;
    }
    if (argc > 4) {
        const size_t ntiles = n / (64 * 16);
#define PP(MODE, RM, G) t = timeit([&] { k_copy_pipe<16, MODE, RM><<<G, 256>>>(a, o, ntiles); }); \
    printf("copy pipelined U=16 mode=%d remap=%d grid=%d: %.2f TB/s\n", MODE, (int)RM, G, 2.0 * bytes / t / 1e12);
        PP(0, false, 1024) PP(0, false, 2048) PP(0, true, 1024) PP(0, true, 2048) PP(1, false, 1024) PP(1, true, 1024) PP(0, false, 512) PP(0, false, 4096) PP(0, true, 4096) PP(1, true, 2048)
        t = timeit([&] { k_copy_tile_nt<16, true, true, 256><<<(unsigned)(n / (16 * 256)), 256>>>(a, o, n); });
        printf("copy tile U=16 nt-both (one tile per wave): %.2f TB/s\n", 2.0 * bytes / t / 1e12);
        t = timeit([&] { k_copy_tile_xcd<16, true, 256, 0><<<(unsigned)(n / (16 * 256)), 256>>>(a, o, n); });
        printf("copy tile U=16 nt-both xcd-remap: %.2f TB/s\n", 2.0 * bytes / t / 1e12);
        t = timeit([&] { k_copy_tile_nt<1, true, true, 256><<<(unsigned)(n / (1 * 256)), 256>>>(a, o, n); });
        printf("copy tile U=1 nt-both: %.2f TB/s\n", 2.0 * bytes / t / 1e12);
        return 0;
    }
    if (argc > 3) {
#define FL(U, NT, BS, RM) t = timeit([&] { k_fill_tile<U, NT, BS, RM><<<(unsigned)(n / ((size_t)U * BS)), BS>>>(o, n); }); \
    printf("fill tile U=%d nt=%d BS=%d remap=%d: %.2f TB/s\n", U, (int)NT, BS, (int)RM, 1.0 * bytes / t / 1e12);
        FL(16, true, 256, true) FL(16, true, 256, false) FL(16, false, 256, true) FL(16, false, 256, false)
        FL(4, true, 256, false) FL(1, true, 256, false) FL(1, false, 256, false) FL(16, true, 64, false) FL(16, true, 1024, true)
        return 0;
    }
    if (argc > 2) {
#define XC(U, NT, BS, CH) t = timeit([&] { k_copy_tile_xcd<U, NT, BS, CH><<<(unsigned)(n / ((size_t)U * BS)), BS>>>(a, o, n); }); \
    printf("copy tile xcd-remap U=%d nt=%d BS=%d chunk=%d: %.2f TB/s\n", U, (int)NT, BS, CH, 2.0 * bytes / t / 1e12);
        XC(16, true, 256, 0) XC(16, true, 256, 1) XC(16, true, 256, 2) XC(16, true, 256, 4) XC(16, true, 256, 16) XC(16, true, 256, 64)
        XC(16, true, 64, 0) XC(16, true, 64, 4) XC(16, true, 64, 16) XC(16, true, 512, 0) XC(16, true, 512, 4) XC(16, true, 1024, 0)
        XC(4, true, 256, 0) XC(4, true, 256, 4) XC(4, true, 256, 16) XC(1, true, 256, 0) XC(1, true, 256, 16) XC(16, false, 256, 0) XC(16, false, 256, 4)
        t = timeit([&] { k_copy_tile_nt<16, true, true, 256><<<(unsigned)(n / (16 * 256)), 256>>>(a, o, n); });
        printf("copy tile U=16 nt-both (no remap): %.2f TB/s\n", 2.0 * bytes / t / 1e12);
        return 0;
    }
    for (int g : {2048, 8192, 65536}) {
        t = timeit([&] { k_copy<<<g, 256>>>(a, o, n); });
        printf("copy grid-stride grid=%d: %.2f TB/s\n", g, 2.0 * bytes / t / 1e12);
    }
    t = timeit([&] { k_copy_tile<4><<<(unsigned)(n / (4 * 256)), 256>>>(a, o, n); });
    printf("copy tile U=4  (1 KB/lane-group of 64: K=1024 row per wave): %.2f TB/s\n", 2.0 * bytes / t / 1e12);
    t = timeit([&] { k_copy_tile<16><<<(unsigned)(n / (16 * 256)), 256>>>(a, o, n); });
    printf("copy tile U=16 (K=4096 row per wave): %.2f TB/s\n", 2.0 * bytes / t / 1e12);
    t = timeit([&] { k_2r1w_tile<4><<<(unsigned)(n / (4 * 256)), 256>>>(a, b, o, n); });
    printf("2R1W tile U=4 : %.2f TB/s\n", 3.0 * bytes / t / 1e12);
    t = timeit([&] { k_2r1w_tile<16><<<(unsigned)(n / (16 * 256)), 256>>>(a, b, o, n); });
    printf("2R1W tile U=16: %.2f TB/s\n", 3.0 * bytes / t / 1e12);
    t = timeit([&] { k_copy_tile_nt<16, true, false, 256><<<(unsigned)(n / (16 * 256)), 256>>>(a, o, n); });
    printf("copy tile U=16 nt-load : %.2f TB/s\n", 2.0 * bytes / t / 1e12);
    t = timeit([&] { k_copy_tile_nt<16, false, true, 256><<<(unsigned)(n / (16 * 256)), 256>>>(a, o, n); });
    printf("copy tile U=16 nt-store: %.2f TB/s\n", 2.0 * bytes / t / 1e12);
    t = timeit([&] { k_copy_tile_nt<16, true, true, 256><<<(unsigned)(n / (16 * 256)), 256>>>(a, o, n); });
    printf("copy tile U=16 nt-both : %.2f TB/s\n", 2.0 * bytes / t / 1e12);
    t = timeit([&] { k_copy_tile_nt<4, true, true, 256><<<(unsigned)(n / (4 * 256)), 256>>>(a, o, n); });
    printf("copy tile U=4  nt-both : %.2f TB/s\n", 2.0 * bytes / t / 1e12);
    t = timeit([&] { k_copy_tile_nt<4, true, true, 1024><<<(unsigned)(n / (4 * 1024)), 1024>>>(a, o, n); });
    printf("copy tile U=4  nt-both BS=1024: %.2f TB/s\n", 2.0 * bytes / t / 1e12);
    t = timeit([&] { k_copy_tile_nt<1, true, true, 256><<<(unsigned)(n / (1 * 256)), 256>>>(a, o, n); });
    printf("copy tile U=1  nt-both : %.2f TB/s\n", 2.0 * bytes / t / 1e12);
    t = timeit([&] { k_copy_tile_nt<1, false, false, 256><<<(unsigned)(n / (1 * 256)), 256>>>(a, o, n); });
    printf("copy tile U=1  plain   : %.2f TB/s\n", 2.0 * bytes / t / 1e12);
    t = timeit([&] { k_copy_tile_nt<2, false, false, 256><<<(unsigned)(n / (2 * 256)), 256>>>(a, o, n); });
    printf("copy tile U=2  plain   : %.2f TB/s\n", 2.0 * bytes / t / 1e12);
    for (int ldsk : {0, 20, 26, 40, 80}) {   // KB of dynamic LDS per block -> caps blocks per CU at 160/ldsk
        t = timeit([&] { k_copy_tile_nt<16, true, true, 256><<<(unsigned)(n / (16 * 256)), 256, ldsk * 1024>>>(a, o, n); });
        printf("copy tile U=16 nt-both BS=256 lds=%dKB: %.2f TB/s\n", ldsk, 2.0 * bytes / t / 1e12);
        t = timeit([&] { k_copy_tile_nt<16, true, true, 64><<<(unsigned)(n / (16 * 64)), 64, ldsk * 256>>>(a, o, n); });
        printf("copy tile U=16 nt-both BS=64  lds=%dKB: %.2f TB/s\n", ldsk / 4, 2.0 * bytes / t / 1e12);
        t = timeit([&] { k_copy_tile_nt<4, true, true, 256><<<(unsigned)(n / (4 * 256)), 256, ldsk * 1024>>>(a, o, n); });
        printf("copy tile U=4  nt-both BS=256 lds=%dKB: %.2f TB/s\n", ldsk, 2.0 * bytes / t / 1e12);
    }
#define BT(U, BS, NT) t = timeit([&] { k_copy_btile<U, BS, NT><<<(unsigned)(n / ((size_t)U * BS)), BS>>>(a, o, n); }); \
    printf("copy block-tile U=%d BS=%d nt=%d: %.2f TB/s\n", U, BS, (int)NT, 2.0 * bytes / t / 1e12);
    BT(4, 256, false) BT(4, 256, true) BT(16, 256, true) BT(4, 1024, true) BT(16, 1024, true) BT(8, 512, true) BT(2, 256, true) BT(1, 256, true) BT(1, 1024, true) BT(1, 64, true) BT(4, 64, true)
#define BT3(U, BS, NT) t = timeit([&] { k_2r1w_btile<U, BS, NT><<<(unsigned)(n / ((size_t)U * BS)), BS>>>(a, b, o, n); }); \
    printf("2R1W block-tile U=%d BS=%d nt=%d: %.2f TB/s\n", U, BS, (int)NT, 3.0 * bytes / t / 1e12);
    BT3(4, 256, false) BT3(4, 256, true) BT3(16, 64, true) BT3(16, 256, true) BT3(1, 256, true) BT3(4, 64, true) BT3(16, 1024, true)
    t = timeit([&] { k_read<<<8192, 256>>>(a, (float*)o, n); });
    printf("read only : %.2f TB/s\n", 1.0 * bytes / t / 1e12);
    t = timeit([&] { k_write<<<8192, 256>>>(o, n); });
    printf("write only: %.2f TB/s\n", 1.0 * bytes / t / 1e12);
    return 0;
}
