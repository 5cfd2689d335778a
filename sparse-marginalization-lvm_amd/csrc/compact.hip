// Kernel 2: non-zero-support stream compaction and the probability-weighted loss reduction.
//
// Replaces the boolean-mask compaction of lvmhelpers/structmarg.py:113-126 (row-major order of
// torch.nonzero(p > 0)) and the p @ loss contractions of lvmhelpers/marg.py:125,
// lvmhelpers/structmarg.py:140,242.
//
// All kernels are HBM streaming work: one wave per row, 16-byte loads per lane, ballot + mbcnt for
// the in-row ordered positions; 128-byte groups with no positive entry are skipped wave-uniformly.
#include "common.hpp"

#include <mutex>

namespace smlvm {

constexpr int kScanItemsPerBlock = 2048;  // 256 threads x 8 items

__device__ __forceinline__ int lanes_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// ------------------------------------------------------------------------------- count
// lists (optional): the forward's support lists (include/sparsemarg.h, smlvm_sparsemax_fwd_cols);
// a row that has one is counted / compacted from its <= cap entries without touching its K
// probabilities, a row marked -2 is scanned as before.
// one wave counts the positive entries of one row
template <bool VEC>
__device__ __forceinline__ int count_row_scan(const float* __restrict__ pr, int K, int lane)
{
    int c = 0;
    if constexpr (VEC) {
        const float4* p4 = reinterpret_cast<const float4*>(pr);
        const int K4 = K >> 2;
        for (int f = lane; f < K4; f += 64) {
            float4 v = p4[f];
            c += (v.x > 0.0f) + (v.y > 0.0f) + (v.z > 0.0f) + (v.w > 0.0f);
        }
    } else {
        for (int j = lane; j < K; j += 64) c += pr[j] > 0.0f;
    }
    return group_reduce<64>(c, OpSum{});
}

template <bool VEC>
__global__ __launch_bounds__(256) void support_count_kernel(const float* __restrict__ p,
                                                            const int32_t* __restrict__ lists, int cap,
                                                            int32_t* __restrict__ nnz, int64_t R,
                                                            int K)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    if (lists) {
        const int cl = lane < cap ? lists[row * cap + lane] : -1;
        if (__builtin_amdgcn_readfirstlane(cl) >= -1) {   // wave-uniform; -2 / -3: no list
            const int n = __popcll(__ballot(cl >= 0));
            if (lane == 0) nnz[row] = n;
            return;
        }
    }
    const int c = count_row_scan<VEC>(p + row * K, K, lane);
    if (lane == 0) nnz[row] = c;
}

// Rows of at most 256 entries (one 16-byte load per lane covers a row): a wave takes four
// consecutive rows and issues their loads together -- with one row per wave the launch holds too
// few bytes in flight to cover the HBM latency (4.3 TB/s at K = 256).
__global__ __launch_bounds__(256) void support_count_short_kernel(const float* __restrict__ p,
                                                                  int32_t* __restrict__ nnz,
                                                                  int64_t R, int K)
{
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (row0 >= R) return;
    const int K4 = K >> 2;
    float4 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < K4 && row0 + r < R) v[r] = reinterpret_cast<const float4*>(p + (row0 + r) * K)[lane];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int c = (v[r].x > 0.0f) + (v[r].y > 0.0f) + (v[r].z > 0.0f) + (v[r].w > 0.0f);
        c = group_reduce<64>(c, OpSum{});
        if (lane == 0 && row0 + r < R) nnz[row0 + r] = c;
    }
}

// sum over groups of G consecutive lanes, G in {4, 8, 16}
template <int G>
__device__ __forceinline__ int lanes_sum(int v)
{
    v += dpp_i32<DPP_QUAD_XOR1>(v);
    v += dpp_i32<DPP_QUAD_XOR2>(v);
    if constexpr (G >= 8) v += dpp_i32<DPP_ROW_HALF_MIRROR>(v);
    if constexpr (G >= 16) v += dpp_i32<DPP_ROW_MIRROR>(v);
    return v;
}

// List-fed count, cap = 4 G: a group of G lanes reads one row's list as 16-byte quarters, so a
// wave takes 64 / G rows with one load (a wave per 128-byte list made the launch latency bound: 1M
// waves for 134 MB).  Rows without a list (first entry -2 / -3) are scanned afterwards, one at a
// time by the whole wave.
template <int G, bool VEC>
__global__ __launch_bounds__(256) void support_count_lists_kernel(const float* __restrict__ p,
                                                                  const int32_t* __restrict__ lists,
                                                                  int32_t* __restrict__ nnz,
                                                                  int64_t R, int K)
{
    constexpr int RW = 64 / G;
    const int lane = threadIdx.x & 63, gl = lane % G;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
    if (row0 >= R) return;
    const int64_t row = row0 + lane / G;
    int4 q = make_int4(-1, -1, -1, -1);
    if (row < R) q = reinterpret_cast<const int4*>(lists + row * (4 * G))[gl];
    const int n = lanes_sum<G>((q.x >= 0) + (q.y >= 0) + (q.z >= 0) + (q.w >= 0));
    const bool unlisted = gl == 0 && row < R && q.x < -1;
    if (gl == 0 && row < R && !unlisted) nnz[row] = n;
    unsigned long long todo = __ballot(unlisted);   // wave-uniform from here on
    if (VEC && K <= 256) {   // short rows: four at a time, loads issued together
        const int K4 = K >> 2;
        while (todo != 0ull) {
            int64_t rr[4];
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rr[j] = -1;
                if (todo != 0ull) { rr[j] = row0 + __builtin_ctzll(todo) / G; todo &= todo - 1ull; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rr[j] >= 0 && lane < K4) v[j] = reinterpret_cast<const float4*>(p + rr[j] * K)[lane];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (rr[j] < 0) break;
                int c = (v[j].x > 0.0f) + (v[j].y > 0.0f) + (v[j].z > 0.0f) + (v[j].w > 0.0f);
                c = group_reduce<64>(c, OpSum{});
                if (lane == 0) nnz[rr[j]] = c;
            }
        }
        return;
    }
    for (; todo != 0ull; todo &= todo - 1ull) {
        const int64_t r = row0 + (__builtin_ctzll(todo)) / G;
        const int c = count_row_scan<VEC>(p + r * K, K, lane);
        if (lane == 0) nnz[r] = c;
    }
}

// ------------------------------------------------------------------------------- offsets
// Three-phase exclusive scan of nnz[R] -> row_offsets[R+1] (int64), block sums in the workspace.
__device__ __forceinline__ long long block_exclusive_scan(long long v, long long* lds,
                                                          long long& total)
{
    // inclusive scan inside the wave with shuffles, then across the 4 waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        long long o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    long long base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        long long s = lds[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(256) void scan_block_sums_kernel(const int32_t* __restrict__ nnz,
                                                              long long* __restrict__ block_sums,
                                                              int64_t R)
{
    const int64_t base = (int64_t)blockIdx.x * kScanItemsPerBlock;
    long long s = 0;
#pragma unroll
    for (int i = 0; i < kScanItemsPerBlock / 256; ++i) {
        int64_t idx = base + i * 256 + threadIdx.x;
        if (idx < R) s += nnz[idx];
    }
    s = group_reduce<64>((double)s, OpSum{});  // exact: < 2^53
    __shared__ long long lds[4];
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ __launch_bounds__(256) void scan_of_sums_kernel(long long* __restrict__ block_sums,
                                                           int64_t nblocks)
{
    __shared__ long long lds[4];
    long long carry = 0;
    for (int64_t start = 0; start < nblocks; start += 256) {
        int64_t idx = start + threadIdx.x;
        long long v = idx < nblocks ? block_sums[idx] : 0;
        long long total;
        long long ex = block_exclusive_scan(v, lds, total);
        if (idx < nblocks) block_sums[idx] = carry + ex;
        carry += total;
    }
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const int32_t* __restrict__ nnz,
                                                         const long long* __restrict__ block_sums,
                                                         int64_t* __restrict__ row_offsets,
                                                         int64_t R)
{
    __shared__ long long lds[4];
    const int64_t base = (int64_t)blockIdx.x * kScanItemsPerBlock;
    // thread t owns 8 consecutive items so the per-thread serial part stays in order
    const int64_t first = base + (int64_t)threadIdx.x * 8;
    long long v[8], s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = (first + i < R) ? nnz[first + i] : 0;
        s += v[i];
    }
    long long total;
    long long ex = block_exclusive_scan(s, lds, total) + block_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (first + i < R) row_offsets[first + i] = ex;
        ex += v[i];
        if (first + i == R - 1) row_offsets[R] = ex;
    }
}

// ------------------------------------------------------------------------------- compact
// one 64-lane step of the VEC scan: `v` is this lane's float4 with index f of the row; the
// positives go out in (lane, component) order from position `out` on (advanced by their number)
__device__ __forceinline__ void compact_step4(float4 v, int f, int64_t row, int64_t& out,
                                              int32_t* __restrict__ rows, int32_t* __restrict__ cols,
                                              float* __restrict__ vals)
{
    const float e[4] = {v.x, v.y, v.z, v.w};
    // positives before this lane's first one: four ballots + mbcnt instead of a shuffle scan
    unsigned long long m[4];
    int pre = 0, total = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        m[c] = __ballot(e[c] > 0.0f);
        pre += lanes_below(m[c]);
        total += __popcll(m[c]);
    }
    if (total == 0) return;  // wave-uniform: nothing in these 1 KiB
    int64_t pos = out + pre;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (e[c] > 0.0f) {
            if (rows) rows[pos] = (int32_t)row;
            if (cols) cols[pos] = f * 4 + c;
            if (vals) vals[pos] = e[c];
            ++pos;
        }
    }
    out += total;
}

// one wave compacts one row, entries in column order from position `out` on
template <bool VEC>
__device__ __forceinline__ void compact_row_scan(const float* __restrict__ pr, int K, int lane,
                                                 int64_t row, int64_t out,
                                                 int32_t* __restrict__ rows, int32_t* __restrict__ cols,
                                                 float* __restrict__ vals)
{
    if constexpr (VEC) {
        const float4* p4 = reinterpret_cast<const float4*>(pr);
        const int K4 = K >> 2;
        for (int f0 = 0; f0 < K4; f0 += 64) {
            const int f = f0 + lane;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < K4) v = p4[f];
            compact_step4(v, f, row, out, rows, cols, vals);
        }
    } else {
        for (int j0 = 0; j0 < K; j0 += 64) {
            const int j = j0 + lane;
            const float v = j < K ? pr[j] : 0.0f;
            const unsigned long long m = __ballot(v > 0.0f);
            if (m == 0ull) continue;
            if (v > 0.0f) {
                const int64_t pos = out + lanes_below(m);
                if (rows) rows[pos] = (int32_t)row;
                if (cols) cols[pos] = j;
                if (vals) vals[pos] = v;
            }
            out += __popcll(m);
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void support_compact_kernel(
    const float* __restrict__ p, const int32_t* __restrict__ lists, int cap,
    const int64_t* __restrict__ row_offsets, int32_t* __restrict__ rows,
    int32_t* __restrict__ cols, float* __restrict__ vals, int64_t R, int K)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* pr = p + row * K;
    const int64_t out = row_offsets[row];
    if (lists) {
        const int cl = lane < cap ? lists[row * cap + lane] : -1;
        if (__builtin_amdgcn_readfirstlane(cl) >= -1) {   // wave-uniform; -2 / -3: no list
            // the list is in no particular order: a lane's output slot is the rank of its column
            const int n = __popcll(__ballot(cl >= 0));
            int rank = 0;
            for (int k = 0; k < n; ++k) rank += __builtin_amdgcn_readlane(cl, k) < cl ? 1 : 0;
            if (cl >= 0) {
                const int64_t pos = out + rank;
                if (rows) rows[pos] = (int32_t)row;
                if (cols) cols[pos] = cl;
                if (vals) vals[pos] = pr[cl];
            }
            return;
        }
    }
    compact_row_scan<VEC>(pr, K, lane, row, out, rows, cols, vals);
}

// Rows of at most 256 entries: four consecutive rows per wave, their loads issued together (see
// support_count_short_kernel).
__global__ __launch_bounds__(256) void support_compact_short_kernel(
    const float* __restrict__ p, const int64_t* __restrict__ row_offsets, int32_t* __restrict__ rows,
    int32_t* __restrict__ cols, float* __restrict__ vals, int64_t R, int K)
{
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (row0 >= R) return;
    const int K4 = K >> 2;
    float4 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < K4 && row0 + r < R) v[r] = reinterpret_cast<const float4*>(p + (row0 + r) * K)[lane];
    }
    // rows are consecutive, so are their output ranges: one offset load, then running positions
    int64_t out = row_offsets[row0];
#pragma unroll
    for (int r = 0; r < 4; ++r) compact_step4(v[r], lane, row0 + r, out, rows, cols, vals);
}

// List-fed compaction, cap = 4 G (see support_count_lists_kernel): 64 / G rows per wave.  The
// lists are in no particular order, so an entry's output slot is the rank of its column inside its
// row: the wave parks its lists in LDS and every lane compares its four entries with the first
// n_max entries of its row (n_max = longest list in the wave; entries are packed at the front).
template <int G, bool VEC>
__global__ __launch_bounds__(256) void support_compact_lists_kernel(
    const float* __restrict__ p, const int32_t* __restrict__ lists,
    const int64_t* __restrict__ row_offsets, int32_t* __restrict__ rows,
    int32_t* __restrict__ cols, float* __restrict__ vals, int64_t R, int K)
{
    constexpr int RW = 64 / G, CAP = 4 * G;
    __shared__ int parked[4][64 * 4];
    const int lane = threadIdx.x & 63, gl = lane % G, grp = lane / G, wave = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * RW;
    if (row0 >= R) return;
    const int64_t row = row0 + grp;
    int4 q = make_int4(-1, -1, -1, -1);
    int64_t out = 0;
    if (row < R) {
        q = reinterpret_cast<const int4*>(lists + row * CAP)[gl];
        out = row_offsets[row];
    }
    const bool unlisted = __shfl(q.x, grp * G) < -1;    // the row's first entry, in all its lanes
    const int c[4] = {q.x, q.y, q.z, q.w};
    int n = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) n += (!unlisted && c[i] >= 0) ? 1 : 0;
    n = lanes_sum<G>(n);
    int n_max = n;
    n_max = group_reduce<64>(n_max, OpMaxI{});
    int* mine = parked[wave] + grp * CAP;
    reinterpret_cast<int4*>(mine)[gl] = q;
    wave_lds_sync();
    int rank[4] = {0, 0, 0, 0};
    for (int k = 0; k < n_max; ++k) {   // wave-uniform trip count
        // unsigned compare: the -1 padding of a shorter list is larger than every column
        const unsigned other = (unsigned)mine[k];
#pragma unroll
        for (int i = 0; i < 4; ++i) rank[i] += other < (unsigned)c[i] ? 1 : 0;
    }
    if (!unlisted) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (c[i] >= 0) {
                const int64_t pos = out + rank[i];
                if (rows) rows[pos] = (int32_t)row;
                if (cols) cols[pos] = c[i];
                if (vals) vals[pos] = p[row * K + c[i]];
            }
        }
    }
    unsigned long long todo = __ballot(unlisted && gl == 0 && row < R);   // wave-uniform from here on
    if (VEC && K <= 256) {   // short rows: four at a time, loads issued together
        const int K4 = K >> 2;
        while (todo != 0ull) {
            int64_t rr[4], o[4];
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rr[j] = -1;
                if (todo != 0ull) { rr[j] = row0 + __builtin_ctzll(todo) / G; todo &= todo - 1ull; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                o[j] = 0;
                if (rr[j] >= 0) {
                    o[j] = row_offsets[rr[j]];
                    if (lane < K4) v[j] = reinterpret_cast<const float4*>(p + rr[j] * K)[lane];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (rr[j] < 0) break;
                compact_step4(v[j], lane, rr[j], o[j], rows, cols, vals);
            }
        }
        return;
    }
    for (; todo != 0ull; todo &= todo - 1ull) {
        const int64_t r = row0 + __builtin_ctzll(todo) / G;
        compact_row_scan<VEC>(p + r * K, K, lane, r, row_offsets[r], rows, cols, vals);
    }
}

// ------------------------------------------------------------------------------- weighted reduce
// Fixed-shape, order-deterministic two-stage sum: stage 1 gives each of up to kRedBlocks blocks a
// contiguous slice of the N products (fp64), stage 2 folds the partials in index order.
constexpr int kRedBlocks = 1024;

__global__ __launch_bounds__(256) void wlr_rows_kernel(const float* __restrict__ vals,
                                                       const float* __restrict__ comps,
                                                       const int64_t* __restrict__ row_offsets,
                                                       float* __restrict__ row_out, int64_t R)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    double acc = 0.0;
    for (int64_t j = row_offsets[r]; j < row_offsets[r + 1]; ++j)
        acc += (double)vals[j] * (double)comps[j];
    row_out[r] = (float)acc;
}

__global__ __launch_bounds__(256) void wlr_partial_kernel(const float* __restrict__ vals,
                                                          const float* __restrict__ comps,
                                                          double* __restrict__ partials, int64_t N,
                                                          int64_t per_block)
{
    __shared__ double lds[4];
    const int64_t lo = (int64_t)blockIdx.x * per_block;
    const int64_t hi = lo + per_block < N ? lo + per_block : N;
    double acc = 0.0;
    for (int64_t j = lo + threadIdx.x; j < hi; j += 256) acc += (double)vals[j] * (double)comps[j];
    acc = group_reduce<64>(acc, OpSum{});
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

__global__ __launch_bounds__(64) void wlr_final_kernel(const double* __restrict__ partials,
                                                       int nparts, float inv_b,
                                                       float* __restrict__ total)
{
    double acc = 0.0;
    for (int j = threadIdx.x; j < nparts; j += 64) acc += partials[j];
    acc = group_reduce<64>(acc, OpSum{});
    if (threadIdx.x == 0) total[0] = (float)(acc * (double)inv_b);
}

__global__ __launch_bounds__(256) void wlr_bwd_kernel(const float* __restrict__ vals,
                                                      const float* __restrict__ comps,
                                                      const float* __restrict__ gtotal, float inv_b,
                                                      float* __restrict__ dvals,
                                                      float* __restrict__ dcomps, int64_t N)
{
    const float g = gtotal[0] * inv_b;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < N; j += (int64_t)gridDim.x * 256) {
        if (dvals) dvals[j] = g * comps[j];
        if (dcomps) dcomps[j] = g * vals[j];
    }
}

// ------------------------------------------------------------------------------- unpack
__global__ __launch_bounds__(256) void unpack_configs_kernel(const uint32_t* __restrict__ aset_bits,
                                                             const int32_t* __restrict__ rows,
                                                             const int32_t* __restrict__ cols,
                                                             float* __restrict__ out, int64_t N,
                                                             int n, int cap, int W)
{
    const int64_t total = N * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * 256) {
        const int64_t j = i / n;
        const int k = (int)(i - j * n);
        const uint32_t w = aset_bits[((int64_t)rows[j] * cap + cols[j]) * W + (k >> 5)];
        out[i] = (float)((w >> (k & 31)) & 1u);
    }
}

}  // namespace smlvm

using namespace smlvm;

static int support_count_go(const float* p, const int32_t* lists, int32_t cap, int32_t* nnz,
                            int64_t R, int32_t K, void* stream)
{
    if (R < 0 || K < 1) return SMLVM_ERR_INVALID;
    if (lists && (cap < 1 || cap > 64)) return SMLVM_ERR_INVALID;
    if (R == 0) return SMLVM_OK;
    if (!p || !nnz) return SMLVM_ERR_INVALID;
    const unsigned grid = (unsigned)((R + 3) / 4);
    const bool vec = K % 4 == 0 && aligned16(p);
    hipStream_t st = (hipStream_t)stream;
    auto grid_for = [&](int rows_per_wave) { return dim3((unsigned)((R + 4 * rows_per_wave - 1) / (4 * rows_per_wave))); };
    if (lists && aligned16(lists) && (cap == 16 || cap == 32 || cap == 64)) {
#define SMLVM_COUNT_LISTS(G)                                                                                     \
    if (vec) hipLaunchKernelGGL((support_count_lists_kernel<G, true>), grid_for(64 / G), dim3(256), 0, st, p, lists, nnz, R, K); \
    else hipLaunchKernelGGL((support_count_lists_kernel<G, false>), grid_for(64 / G), dim3(256), 0, st, p, lists, nnz, R, K)
        if (cap == 16) { SMLVM_COUNT_LISTS(4); }
        else if (cap == 32) { SMLVM_COUNT_LISTS(8); }
        else { SMLVM_COUNT_LISTS(16); }
#undef SMLVM_COUNT_LISTS
    } else if (!lists && vec && K <= 256)
        hipLaunchKernelGGL(support_count_short_kernel, grid_for(4), dim3(256), 0, st, p, nnz, R, K);
    else if (vec)
        hipLaunchKernelGGL(support_count_kernel<true>, dim3(grid), dim3(256), 0, st, p, lists, cap, nnz, R, K);
    else
        hipLaunchKernelGGL(support_count_kernel<false>, dim3(grid), dim3(256), 0, st, p, lists, cap, nnz, R, K);
    return check_launch();
}

extern "C" int smlvm_support_count(const float* p, int32_t* nnz, int64_t R, int32_t K, void* stream)
{
    return support_count_go(p, nullptr, 0, nnz, R, K, stream);
}

extern "C" int smlvm_support_count_cols(const float* p, const int32_t* support_cols, int32_t cap,
                                        int32_t* nnz, int64_t R, int32_t K, void* stream)
{
    return support_count_go(p, support_cols, cap, nnz, R, K, stream);
}

extern "C" size_t smlvm_support_offsets_workspace_bytes(int64_t R)
{
    if (R <= 0) return 8;
    return (size_t)((R + kScanItemsPerBlock - 1) / kScanItemsPerBlock) * sizeof(long long);
}

extern "C" int smlvm_support_offsets(const int32_t* nnz, int64_t* row_offsets, int64_t R,
                                     void* workspace, size_t workspace_bytes, void* stream)
{
    if (R < 0 || !row_offsets) return SMLVM_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (R == 0) {
        if (hipMemsetAsync(row_offsets, 0, sizeof(int64_t), st) != hipSuccess) return SMLVM_ERR_HIP;
        return SMLVM_OK;
    }
    if (!nnz) return SMLVM_ERR_INVALID;
    if (!workspace || workspace_bytes < smlvm_support_offsets_workspace_bytes(R))
        return SMLVM_ERR_WORKSPACE;
    const int64_t nblocks = (R + kScanItemsPerBlock - 1) / kScanItemsPerBlock;
    long long* sums = (long long*)workspace;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, nnz, sums, R);
    hipLaunchKernelGGL(scan_of_sums_kernel, dim3(1), dim3(256), 0, st, sums, nblocks);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, nnz, sums, row_offsets, R);
    return check_launch();
}

// ------------------------------------------------------------------------------- no-allocation
// smlvm_support_offsets_noalloc: the same three launches with the block sums in a file-scope device
// array instead of a caller workspace, for callers that cannot allocate (HIP graph capture, tight
// serving loops).  `scratch` below is a HOST-SIDE STAND-IN for the device array, not a
// device address: its address on the current device comes from hipGetSymbolAddress, once per
// device, and only that pointer is ever handed to a kernel.
constexpr int64_t kNoAllocMaxRows = (int64_t)SMLVM_NOALLOC_MAX_ROWS;
constexpr int kNoAllocSums = (int)(kNoAllocMaxRows / kScanItemsPerBlock);   // 8192 block sums = 64 KiB
static __device__ __align__(256) long long scratch[kNoAllocSums];

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static long long* g_scratch_ptr[kMaxDevices];   // per device: each GPU holds its own copy of the array

static long long* offsets_scratch_on_current_device()
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            smlvm::g_last_hip_error = (int)e;
            return nullptr;
        }
        g_scratch_ptr[dev] = (long long*)ptr;
    }
    return g_scratch_ptr[dev];
}

extern "C" int smlvm_support_offsets_noalloc(const int32_t* nnz, int64_t* row_offsets, int64_t R,
                                             void* stream)
{
    if (R < 0 || !row_offsets) return SMLVM_ERR_INVALID;
    if (R > kNoAllocMaxRows) return SMLVM_ERR_UNSUPPORTED;   // beyond the static block-sum array
    hipStream_t st = (hipStream_t)stream;
    if (R == 0) {
        if (hipMemsetAsync(row_offsets, 0, sizeof(int64_t), st) != hipSuccess) return SMLVM_ERR_HIP;
        return SMLVM_OK;
    }
    if (!nnz) return SMLVM_ERR_INVALID;
    long long* sums = offsets_scratch_on_current_device();
    if (!sums) return SMLVM_ERR_HIP;
    // every block sum the later launches read is written by the first one: nothing has to be cleared
    const int64_t nblocks = (R + kScanItemsPerBlock - 1) / kScanItemsPerBlock;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, nnz, sums, R);
    hipLaunchKernelGGL(scan_of_sums_kernel, dim3(1), dim3(256), 0, st, sums, nblocks);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, nnz, sums, row_offsets, R);
    return check_launch();
}

static int support_compact_go(const float* p, const int32_t* lists, int32_t cap,
                              const int64_t* row_offsets, int32_t* rows, int32_t* cols, float* vals,
                              int64_t R, int32_t K, void* stream)
{
    if (R < 0 || K < 1) return SMLVM_ERR_INVALID;
    if (lists && (cap < 1 || cap > 64)) return SMLVM_ERR_INVALID;
    if (R == 0) return SMLVM_OK;
    if (!p || !row_offsets) return SMLVM_ERR_INVALID;
    const unsigned grid = (unsigned)((R + 3) / 4);
    const bool vec = K % 4 == 0 && aligned16(p);
    hipStream_t st = (hipStream_t)stream;
    auto grid_for = [&](int rows_per_wave) { return dim3((unsigned)((R + 4 * rows_per_wave - 1) / (4 * rows_per_wave))); };
    if (lists && aligned16(lists) && (cap == 16 || cap == 32 || cap == 64)) {
#define SMLVM_COMPACT_LISTS(G)                                                                                   \
    if (vec) hipLaunchKernelGGL((support_compact_lists_kernel<G, true>), grid_for(64 / G), dim3(256), 0, st, p, lists, row_offsets, rows, cols, vals, R, K); \
    else hipLaunchKernelGGL((support_compact_lists_kernel<G, false>), grid_for(64 / G), dim3(256), 0, st, p, lists, row_offsets, rows, cols, vals, R, K)
        if (cap == 16) { SMLVM_COMPACT_LISTS(4); }
        else if (cap == 32) { SMLVM_COMPACT_LISTS(8); }
        else { SMLVM_COMPACT_LISTS(16); }
#undef SMLVM_COMPACT_LISTS
    } else if (!lists && vec && K <= 256)
        hipLaunchKernelGGL(support_compact_short_kernel, grid_for(4), dim3(256), 0, st, p, row_offsets, rows, cols, vals, R, K);
    else if (vec)
        hipLaunchKernelGGL(support_compact_kernel<true>, dim3(grid), dim3(256), 0, st, p, lists, cap, row_offsets, rows, cols, vals, R, K);
    else
        hipLaunchKernelGGL(support_compact_kernel<false>, dim3(grid), dim3(256), 0, st, p, lists, cap, row_offsets, rows, cols, vals, R, K);
    return check_launch();
}

extern "C" int smlvm_support_compact(const float* p, const int64_t* row_offsets, int32_t* rows,
                                     int32_t* cols, float* vals, int64_t R, int32_t K, void* stream)
{
    return support_compact_go(p, nullptr, 0, row_offsets, rows, cols, vals, R, K, stream);
}

extern "C" int smlvm_support_compact_cols(const float* p, const int32_t* support_cols, int32_t cap,
                                          const int64_t* row_offsets, int32_t* rows, int32_t* cols,
                                          float* vals, int64_t R, int32_t K, void* stream)
{
    return support_compact_go(p, support_cols, cap, row_offsets, rows, cols, vals, R, K, stream);
}

extern "C" size_t smlvm_weighted_loss_reduce_workspace_bytes(int64_t N)
{
    (void)N;
    return (size_t)kRedBlocks * sizeof(double);
}

extern "C" int smlvm_weighted_loss_reduce_fwd(const float* vals, const float* comps,
                                              const int64_t* row_offsets, float* row_out,
                                              float* total, float inv_b, int64_t R, int64_t N,
                                              void* workspace, size_t workspace_bytes, void* stream)
{
    if (R < 0 || N < 0) return SMLVM_ERR_INVALID;
    if (N > 0 && (!vals || !comps)) return SMLVM_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (row_out && R > 0) {
        if (!row_offsets) return SMLVM_ERR_INVALID;
        hipLaunchKernelGGL(wlr_rows_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, vals, comps, row_offsets, row_out, R);
    }
    if (total) {
        if (!workspace || workspace_bytes < smlvm_weighted_loss_reduce_workspace_bytes(N))
            return SMLVM_ERR_WORKSPACE;
        double* partials = (double*)workspace;
        int64_t per_block = (N + kRedBlocks - 1) / kRedBlocks;
        if (per_block < 256) per_block = 256;
        int nparts = (int)((N + per_block - 1) / per_block);
        if (nparts > 0)
            hipLaunchKernelGGL(wlr_partial_kernel, dim3((unsigned)nparts), dim3(256), 0, st, vals, comps, partials, N, per_block);
        hipLaunchKernelGGL(wlr_final_kernel, dim3(1), dim3(64), 0, st, partials, nparts, inv_b, total);
    }
    return check_launch();
}

extern "C" int smlvm_weighted_loss_reduce_bwd(const float* vals, const float* comps,
                                              const float* gtotal, float inv_b, float* dvals,
                                              float* dcomps, int64_t N, void* stream)
{
    if (N < 0) return SMLVM_ERR_INVALID;
    if (N == 0) return SMLVM_OK;
    if (!vals || !comps || !gtotal) return SMLVM_ERR_INVALID;
    int64_t blocks = (N + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wlr_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, vals, comps, gtotal, inv_b, dvals, dcomps, N);
    return check_launch();
}

extern "C" int smlvm_unpack_configs(const uint32_t* aset_bits, const int32_t* rows,
                                    const int32_t* cols, float* out, int64_t N, int32_t n,
                                    void* stream)
{
    if (N < 0 || n < 1) return SMLVM_ERR_INVALID;
    if (N == 0) return SMLVM_OK;
    if (!aset_bits || !rows || !cols || !out) return SMLVM_ERR_INVALID;
    int64_t blocks = (N * n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(unpack_configs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, aset_bits, rows, cols, out, N, n, n + 1, (n + 31) / 32);
    return check_launch();
}
