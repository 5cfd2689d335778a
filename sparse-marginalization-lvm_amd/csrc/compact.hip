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

#ifndef SMLVM_FUSED_WAVES
#define SMLVM_FUSED_WAVES 8
#endif

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

// ------------------------------------------------------------------------------- fused single pass
// smlvm_support_compact_fused: count + exclusive scan + compaction in ONE launch (round 3; the
// chain count -> three scan launches -> compact read the lists twice and cost five launches).
// A workgroup of 8 waves takes a tile of 512 consecutive rows: phase A counts every row (from its support
// list, or by scanning p for rows without one), the counts are scanned in the block, the
// tile's total is published and the exclusive prefix of the tiles before it comes from a decoupled
// look-back over their status words (one 64-bit word per tile: 2 flag bits + 62 value bits, so a
// word is consistent by itself and relaxed device-scope atomics suffice); phase B writes the row
// offsets and the compacted entries.  Tiles are handed out by a ticket counter, so every tile a
// look-back waits for has already started and publishes its total without waiting for anyone.
// The wait is bounded all the same (kLookbackSpins polls, then the tile gives up, raises the
// error word of the workspace and every later result of the launch is void): no wave can spin
// forever.  With the forward's value lists (smlvm_sparsemax_fwd_lists) nothing is gathered from p.
constexpr int kFusedWaves = SMLVM_FUSED_WAVES;            // waves per workgroup = 64-row slices per tile
constexpr int kFusedTileRows = 64 * kFusedWaves;
constexpr unsigned long long kTileAggregate = 1ull << 62, kTilePrefix = 2ull << 62, kTileValue = (1ull << 62) - 1ull;
constexpr int kLookbackSpins = 1 << 22;

// workspace: [0] ticket counter, [1] error word, [2] N (copy of row_offsets[R]), [3] reserved,
// [4 ...] one status word per tile; all zero before the launch (the entry point clears it)
struct FusedWs { unsigned long long* w; };

__device__ __forceinline__ void compact_step4_lim(float4 v, int f, int64_t row, int64_t& out, int64_t lim,
                                                  int32_t* __restrict__ rows, int32_t* __restrict__ cols,
                                                  float* __restrict__ vals)
{
    const float e[4] = {v.x, v.y, v.z, v.w};
    unsigned long long m[4];
    int pre = 0, total = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        m[c] = __ballot(e[c] > 0.0f);
        pre += lanes_below(m[c]);
        total += __popcll(m[c]);
    }
    if (total == 0) return;
    int64_t pos = out + pre;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (e[c] > 0.0f) {
            if (pos < lim) {
                if (rows) rows[pos] = (int32_t)row;
                if (cols) cols[pos] = f * 4 + c;
                if (vals) vals[pos] = e[c];
            }
            ++pos;
        }
    }
    out += total;
}

template <bool VEC>
__device__ __forceinline__ void compact_row_scan_lim(const float* __restrict__ pr, int K, int lane,
                                                     int64_t row, int64_t out, int64_t lim,
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
            compact_step4_lim(v, f, row, out, lim, rows, cols, vals);
        }
    } else {
        for (int j0 = 0; j0 < K; j0 += 64) {
            const int j = j0 + lane;
            const float v = j < K ? pr[j] : 0.0f;
            const unsigned long long m = __ballot(v > 0.0f);
            if (m == 0ull) continue;
            if (v > 0.0f) {
                const int64_t pos = out + lanes_below(m);
                if (pos < lim) {
                    if (rows) rows[pos] = (int32_t)row;
                    if (cols) cols[pos] = j;
                    if (vals) vals[pos] = v;
                }
            }
            out += __popcll(m);
        }
    }
}

// G lanes per row (cap = 4 G), S = G steps of 64 / G rows per wave: 64 rows per wave, 64 x
// kFusedWaves per tile.
// lists == nullptr: every row is scanned (G only fixes the shape of the loop then).
// exclusive scan of one value per thread over a workgroup of NW waves (NW <= 16)
template <int NW>
__device__ __forceinline__ long long block_exclusive_scan_nw(long long v, long long* lds, long long& total)
{
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
    for (int w = 0; w < NW; ++w) {
        long long s = lds[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// LR: no lists at all and K <= 64 -- a lane per row (below); a variant of its own, so that the
// list-fed kernel keeps the code it was tuned with (as a run-time branch it cost that one 8 %)
// PAT: rows without a list are COUNTED from the 16-bit-per-lane patterns the forward leaves for
// every row of 32 < K <= 1024 (smlvm_sparsemax_support_mask_bytes: pat_dwords = lanes / 2 words per
// row) instead of from their K probabilities -- a lane per row, one or two 16-byte loads, no
// reduction; the entries themselves are still read from p in phase B.
template <int G, bool VEC, bool LR = false, bool PAT = false>
__global__ __launch_bounds__(64 * kFusedWaves) void support_compact_fused_kernel(
    const float* __restrict__ p, const int32_t* __restrict__ lists, const float* __restrict__ vlists,
    int64_t* __restrict__ row_offsets, int32_t* __restrict__ rows, int32_t* __restrict__ cols,
    float* __restrict__ vals, int64_t n_cap, int64_t R, int K, unsigned long long* __restrict__ ws,
    const uint32_t* __restrict__ patterns = nullptr, int pat_dwords = 0)
{
    constexpr int RW = 64 / G, S = G, CAP = 4 * G;
    __shared__ int cnt[kFusedTileRows];
    __shared__ long long offs[kFusedTileRows];
    __shared__ long long scan_lds[kFusedWaves];
    __shared__ long long tile_info[2];      // [0] ticket, [1] exclusive prefix of the tile
    __shared__ int parked[kFusedWaves][64 * 4];
    // a step's entries are staged here in output order and leave as dense, coalesced stores (each
    // lane storing its own few entries cost 12 sparse store instructions per step: the launch was
    // bound by their issue, not by bytes)
    __shared__ int stage_c[kFusedWaves][64 * 4];
    __shared__ float stage_v[kFusedWaves][64 * 4];
    __shared__ unsigned char stage_g[kFusedWaves][64 * 4];
    const int lane = threadIdx.x & 63, gl = lane % G, grp = lane / G, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        tile_info[0] = (long long)__hip_atomic_fetch_add(&ws[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int64_t tile = tile_info[0];
    const int64_t ntiles = (R + kFusedTileRows - 1) / kFusedTileRows;
    if (tile >= ntiles) return;             // (never: the grid has exactly ntiles blocks)
    const int64_t row_base = tile * kFusedTileRows + wave * 64;

    // ---- phase A: counts
    int4 q[S];
    unsigned long long unl_steps[S];       // per step: ballot of (gl == 0 && the row has no list)
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int64_t row = row_base + s * RW + grp;
        q[s] = make_int4(-1, -1, -1, -1);
        if (lists) { if (row < R) q[s] = reinterpret_cast<const int4*>(lists + row * CAP)[gl]; }
        else q[s].x = -2;
        const int first = __shfl(q[s].x, grp * G);
        const bool unlisted = first < -1;
        const int n = lanes_sum<G>((q[s].x >= 0) + (q[s].y >= 0) + (q[s].z >= 0) + (q[s].w >= 0));
        unl_steps[s] = __ballot(unlisted && gl == 0 && row < R);
        if (gl == 0) cnt[wave * 64 + s * RW + grp] = (row < R && !unlisted) ? n : 0;
        if (unlisted) q[s] = make_int4(-1, -1, -1, -1);
    }
    // without value lists the listed entries' probabilities are gathered from p now, while the
    // counts are scanned and the look-back waits (K = 256: 0.191 -> 0.167 ms; fetching the value
    // lists this early instead of in phase B was measured too: 0.125 -> 0.135 ms, not kept)
    float4 pv[S];
    if (vals && !vlists) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int64_t row = row_base + s * RW + grp;
            pv[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q[s].x >= 0) {
                const float* pr = p + row * K;
                pv[s].x = pr[q[s].x];
                if (q[s].y >= 0) pv[s].y = pr[q[s].y];
                if (q[s].z >= 0) pv[s].z = pr[q[s].z];
                if (q[s].w >= 0) pv[s].w = pr[q[s].w];
            }
        }
    }
    // No lists at all and short rows (the SparseMAP distributions: [B, n + 1] with n <= 63, or a
    // [B, k] top-k distribution): a lane per row.  A wave per row, one row after the other, is 128
    // dependent round trips for a tile of 64 rows -- 47 us for the 64 x 33 batch of the 32-bit VAE,
    // which the host then waits for (it reads the offsets back).
    constexpr bool lane_rows = LR;
    if constexpr (PAT) {
        // (the listed rows' counts are in cnt[] already, 0 for the unlisted ones of this slice)
        unsigned long long unl_any = 0ull;
#pragma unroll
        for (int s = 0; s < S; ++s) unl_any |= unl_steps[s];
        if (unl_any != 0ull) {                 // wave-uniform
            // which rows of the slice are unlisted: slice row i = s * RW + grp has its flag at bit
            // grp * G of unl_steps[s]; parked as one byte per row in this wave's LDS
            unsigned char* flag = reinterpret_cast<unsigned char*>(parked[wave]);
            wave_lds_sync();
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (gl == 0) flag[s * RW + grp] = (unsigned char)((unl_steps[s] >> (grp * G)) & 1ull);
            wave_lds_sync();
            const int64_t r = row_base + lane;
            if (r < R && flag[lane]) {
                const uint4* pr = reinterpret_cast<const uint4*>(patterns + r * pat_dwords);
                int c = 0;
                for (int d = 0; d < (pat_dwords >> 2); ++d) {
                    const uint4 w = pr[d];
                    c += __popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w);
                }
                cnt[wave * 64 + lane] = c;
            }
            wave_lds_sync();
        }
    } else if constexpr (lane_rows) {
        const int64_t r = row_base + lane;
        int c = 0;
        if (r < R) {
            const float* pr = p + r * K;
#pragma unroll 8
            for (int k = 0; k < K; ++k) c += pr[k] > 0.0f ? 1 : 0;
        }
        cnt[wave * 64 + lane] = c;
    } else if (VEC && G <= 8 && K <= 256) {
        // short rows without a list (narrow logits: more non-zeros than a list holds), four at a
        // time with their loads in flight together -- one row after the other, each a dependent
        // round trip, made 1M x 256 rows of N(0, 0.05^2) latency bound (0.65 ms)
        const int K4 = K >> 2;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            for (unsigned long long todo = unl_steps[s]; todo != 0ull;) {   // wave-uniform
                int g[4], n = 0;
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    g[u] = 0;
                    if (todo != 0ull) {
                        g[u] = __builtin_ctzll(todo) / G;
                        todo &= todo - 1ull;
                        n = u + 1;
                        if (lane < K4) v[u] = reinterpret_cast<const float4*>(p + (row_base + s * RW + g[u]) * K)[lane];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u < n) {
                        int c = (v[u].x > 0.0f) + (v[u].y > 0.0f) + (v[u].z > 0.0f) + (v[u].w > 0.0f);
                        c = group_reduce<64>(c, OpSum{});
                        if (lane == 0) cnt[wave * 64 + s * RW + g[u]] = c;
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            for (unsigned long long todo = unl_steps[s]; todo != 0ull; todo &= todo - 1ull) {   // wave-uniform
                const int g = __builtin_ctzll(todo) / G;
                const int64_t r = row_base + s * RW + g;
                const int c = count_row_scan<VEC>(p + r * K, K, lane);
                if (lane == 0) cnt[wave * 64 + s * RW + g] = c;
            }
        }
    }
    __syncthreads();
    long long total;
    const long long mine = cnt[threadIdx.x];
    const long long ex = block_exclusive_scan_nw<kFusedWaves>(mine, scan_lds, total);

    // ---- the tile's exclusive prefix: decoupled look-back (wave 0)
    if (wave == 0) {
        unsigned long long* status = ws + 4;
        if (lane == 0)
            __hip_atomic_store(&status[tile], (tile == 0 ? kTilePrefix : kTileAggregate) | (unsigned long long)total,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long excl = 0;
        if (tile > 0) {
            int64_t look = tile - 1;
            int spins = 0;
            for (;;) {
                const int64_t idx = look - lane;
                unsigned long long w = kTilePrefix;   // before tile 0: an inclusive prefix of zero
                if (idx >= 0) w = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned flag = (unsigned)(w >> 62);
                const unsigned long long m_inv = __ballot(flag == 0u), m_pre = __ballot(flag == 2u);
                const int first_p = m_pre ? __builtin_ctzll(m_pre) : 64;
                const unsigned long long need = first_p >= 63 ? ~0ull : ((2ull << first_p) - 1ull);
                if ((m_inv & need) != 0ull) {         // a tile in the window has not published yet
                    if (++spins > kLookbackSpins) {
                        if (lane == 0) __hip_atomic_store(&ws[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                long long v = (lane <= first_p) ? (long long)(w & kTileValue) : 0ll;
                v = (long long)group_reduce<64>((double)v, OpSum{});   // exact: < 2^53
                excl += v;
                if (first_p < 64) break;
                look -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(&status[tile], kTilePrefix | (unsigned long long)(excl + total),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) tile_info[1] = excl;
    }
    __syncthreads();
    const long long prefix = tile_info[1];

    // ---- phase B: offsets, then the entries
    {
        const int64_t row = tile * kFusedTileRows + threadIdx.x;
        const long long o = prefix + ex;
        offs[threadIdx.x] = o;
        if (row < R) row_offsets[row] = o;
        if (row == R - 1) {
            row_offsets[R] = o + mine;
            __hip_atomic_store(&ws[2], (unsigned long long)(o + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!rows && !cols && !vals) return;
    int* my_park = parked[wave] + grp * CAP;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int64_t row = row_base + s * RW + grp;
        const int c[4] = {q[s].x, q[s].y, q[s].z, q[s].w};
        const bool any = c[0] >= 0;            // entries are packed at the front of a list
        if (__ballot(any) != 0ull) {
            int n = (c[0] >= 0) + (c[1] >= 0) + (c[2] >= 0) + (c[3] >= 0);
            n = lanes_sum<G>(n);
            const int n_max = group_reduce<64>(n, OpMaxI{});
            wave_lds_sync();
            reinterpret_cast<int4*>(my_park)[gl] = q[s];
            wave_lds_sync();
            float4 pvl = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vals && vlists && any) pvl = reinterpret_cast<const float4*>(vlists + row * CAP)[gl];
            int rank[4] = {0, 0, 0, 0};
            for (int k = 0; k < n_max; k += 4) {   // wave-uniform trip count; four entries per LDS read
                const int4 o4 = reinterpret_cast<const int4*>(my_park)[k >> 2];
                const unsigned other[4] = {(unsigned)o4.x, (unsigned)o4.y, (unsigned)o4.z, (unsigned)o4.w};
#pragma unroll
                for (int u = 0; u < 4; ++u)      // -1 padding: larger than every column
#pragma unroll
                    for (int i = 0; i < 4; ++i) rank[i] += other[u] < (unsigned)c[i] ? 1 : 0;
            }
            const long long out = offs[wave * 64 + s * RW + grp];
            const float4 pvs = vlists ? pvl : pv[s];
            const float pvv[4] = {pvs.x, pvs.y, pvs.z, pvs.w};
            if (unl_steps[s] == 0ull) {
                // every row of the step has its list: the step's entries are one contiguous run of
                // the output, from the first row's offset to the next step's
                const long long base = offs[wave * 64 + s * RW];
                const int first_next = wave * 64 + (s + 1) * RW;
                // (the last step of the tile ends at prefix + total)
                const long long end = first_next < kFusedTileRows ? offs[first_next] : prefix + total;
                const int run = (int)(end - base);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (c[i] >= 0) {
                        const int slot = (int)(out - base) + rank[i];
                        stage_c[wave][slot] = c[i];
                        stage_g[wave][slot] = (unsigned char)grp;
                        if (vals) stage_v[wave][slot] = pvv[i];
                    }
                }
                wave_lds_sync();
                for (int e = lane; e < run; e += 64) {
                    const int64_t pos = base + e;
                    if (pos < n_cap) {
                        if (rows) rows[pos] = (int32_t)(row_base + s * RW + stage_g[wave][e]);
                        if (cols) cols[pos] = stage_c[wave][e];
                        if (vals) vals[pos] = stage_v[wave][e];
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t pos = out + rank[i];
                    if (c[i] >= 0 && pos < n_cap) {
                        if (rows) rows[pos] = (int32_t)row;
                        if (cols) cols[pos] = c[i];
                        if (vals) vals[pos] = pvv[i];
                    }
                }
            }
        }
        if constexpr (lane_rows) continue;
        if (VEC && G <= 8 && K <= 256) {   // four short rows at a time, as in phase A
            const int K4 = K >> 2;
            for (unsigned long long todo = unl_steps[s]; todo != 0ull;) {   // wave-uniform
                int g[4], n = 0;
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    g[u] = 0;
                    if (todo != 0ull) {
                        g[u] = __builtin_ctzll(todo) / G;
                        todo &= todo - 1ull;
                        n = u + 1;
                        if (lane < K4) v[u] = reinterpret_cast<const float4*>(p + (row_base + s * RW + g[u]) * K)[lane];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u < n) {
                        int64_t out = offs[wave * 64 + s * RW + g[u]];
                        compact_step4_lim(v[u], lane, row_base + s * RW + g[u], out, n_cap, rows, cols, vals);
                    }
                }
            }
            continue;
        }
        for (unsigned long long todo = unl_steps[s]; todo != 0ull; todo &= todo - 1ull) {   // wave-uniform
            const int g = __builtin_ctzll(todo) / G;
            const int64_t r = row_base + s * RW + g;
            compact_row_scan_lim<VEC>(p + r * K, K, lane, r, offs[wave * 64 + s * RW + g], n_cap, rows, cols, vals);
        }
    }
    if constexpr (lane_rows) {
        const int64_t r = row_base + lane;
        if (r < R) {
            int64_t pos = offs[wave * 64 + lane];
            const float* pr = p + r * K;
#pragma unroll 8
            for (int k = 0; k < K; ++k) {
                const float v = pr[k];
                if (v > 0.0f) {
                    if (pos < n_cap) {
                        if (rows) rows[pos] = (int32_t)r;
                        if (cols) cols[pos] = k;
                        if (vals) vals[pos] = v;
                    }
                    ++pos;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------- weighted reduce
// Fixed-shape, order-deterministic two-stage sum: stage 1 gives each of up to kRedBlocks blocks a
// contiguous slice of the N products (fp64), stage 2 folds the partials in index order.
constexpr int kRedBlocks = 1024;
constexpr int64_t kRedSingle = 16384;   // up to here one block sums everything and writes the total itself

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
                                                          int64_t per_block, float scale,
                                                          float* __restrict__ total)
{
    __shared__ double lds[4];
    const int64_t lo = (int64_t)blockIdx.x * per_block;
    const int64_t hi = lo + per_block < N ? lo + per_block : N;
    double acc = 0.0;
    for (int64_t j = lo + threadIdx.x; j < hi; j += 256) acc += (double)vals[j] * (double)comps[j];
    acc = group_reduce<64>(acc, OpSum{});
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double blk = (lds[0] + lds[1]) + (lds[2] + lds[3]);
        // a single block is the whole sum: the second stage's arithmetic, without its launch
        if (total) total[0] = (float)(blk * (double)scale);
        else partials[blockIdx.x] = blk;
    }
}

// -sum_{v > 0} v log v over the compacted (or dense) distribution, same fixed two-stage shape
// (structmarg.py:51-54, 200: `-(distr @ torch.log(distr))`, entries that are not positive skipped)
__global__ __launch_bounds__(256) void plogp_partial_kernel(const float* __restrict__ vals,
                                                            double* __restrict__ partials, int64_t N,
                                                            int64_t per_block, float scale,
                                                            float* __restrict__ total)
{
    __shared__ double lds[4];
    const int64_t lo = (int64_t)blockIdx.x * per_block;
    const int64_t hi = lo + per_block < N ? lo + per_block : N;
    double acc = 0.0;
    for (int64_t j = lo + threadIdx.x; j < hi; j += 256) {
        const float v = vals[j];
        if (v > 0.0f) acc += (double)(v * logf(v));
    }
    acc = group_reduce<64>(acc, OpSum{});
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double blk = (lds[0] + lds[1]) + (lds[2] + lds[3]);
        // a single block is the whole sum: the second stage's arithmetic, without its launch
        if (total) total[0] = (float)(blk * (double)scale);
        else partials[blockIdx.x] = blk;
    }
}

// d/dv of -scale * sum v log v, times the incoming gradient: -(log v + 1) scale g for v > 0, else 0
__global__ __launch_bounds__(256) void plogp_bwd_kernel(const float* __restrict__ vals,
                                                        const float* __restrict__ gout, float scale,
                                                        float* __restrict__ dvals, int64_t N)
{
    const float g = gout[0] * scale;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < N; j += (int64_t)gridDim.x * 256) {
        const float v = vals[j];
        dvals[j] = v > 0.0f ? -(logf(v) + 1.0f) * g : 0.0f;
    }
}

// out[r, :] = sum of g[j, :] over the row's entries j in [row_offsets[r], row_offsets[r+1]), in the
// order of j: the backward of gathering one [D]-vector per compacted entry out of a per-row table
// (state[rows]: the compaction is row-major, so a row's entries are contiguous).  torch's
// index_select backward adds with float atomics, in an order that changes from run to run; this is
// a fixed-order sum.  A thread owns one column of one row (float4 when D allows), so every load is
// coalesced across the block and nothing is reduced across lanes.
template <bool VEC>
__global__ __launch_bounds__(256) void support_segment_sum_kernel(const float* __restrict__ g,
                                                                  const int64_t* __restrict__ row_offsets,
                                                                  float* __restrict__ out, int64_t R, int64_t D)
{
    const int64_t r = blockIdx.y;
    const int64_t lo = row_offsets[r], hi = row_offsets[r + 1];
    if constexpr (VEC) {
        const int64_t d4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (d4 * 4 >= D) return;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t j = lo; j < hi; ++j) {
            const float4 v = reinterpret_cast<const float4*>(g + j * D)[d4];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        reinterpret_cast<float4*>(out + r * D)[d4] = acc;
    } else {
        const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (d >= D) return;
        float acc = 0.f;
        for (int64_t j = lo; j < hi; ++j) acc += g[j * D + d];
        out[r * D + d] = acc;
    }
}

// out[r, :] = 0 except out[r, cols[j]] = g[j] for the row's entries j in [row_offsets[r],
// row_offsets[r+1]): the backward of gathering the compacted values out of a dense [R, K] tensor
// (positions are distinct, so nothing has to be sorted or accumulated).  One wave per row.
__global__ __launch_bounds__(256) void support_scatter_kernel(const float* __restrict__ g,
                                                              const int64_t* __restrict__ row_offsets,
                                                              const int32_t* __restrict__ cols,
                                                              float* __restrict__ out, int64_t R, int K)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;   // whole wave
    float* o = out + r * K;
    for (int k = lane; k < K; k += 64) o[k] = 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the zeros before the entries, same wave
    __builtin_amdgcn_wave_barrier();
    const int64_t lo = row_offsets[r], hi = row_offsets[r + 1];
    for (int64_t j = lo + lane; j < hi; j += 64) {
        const int c = cols[j];
        if (c >= 0 && c < K) o[c] = g[j];
    }
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

// Zeroes `n` 8-byte words.  A kernel, not hipMemsetAsync: inside a stream capture the memset becomes
// a graph node of its own, and with the HIP runtime inside this image's PyTorch wheel such a node
// writes zero on the first replay only and an arbitrary word afterwards (csrc/graphfix.hip): the
// ticket counter then started above the tile count, every tile of the fused compaction returned
// without writing its offsets, and the scatter behind it read somebody else's bytes as offsets.
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned long long* __restrict__ w, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) w[i] = 0ull;
}

static int zero_words(void* dst, size_t bytes, hipStream_t st)
{
    const long long n = (long long)(bytes / sizeof(unsigned long long));
    if (n == 0) return SMLVM_OK;
    unsigned grid = (unsigned)((n + 255) / 256);
    if (grid > 64u) grid = 64u;
    hipLaunchKernelGGL(zero_words_kernel, dim3(grid), dim3(256), 0, st, static_cast<unsigned long long*>(dst), n);
    return SMLVM_OK;
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
        zero_words(row_offsets, sizeof(int64_t), st);
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

// The array of the GPU the launches will run on: the stream's device when the stream knows it
// (a stream of another GPU than the current one must not get the current GPU's array), the
// current device for the null stream.  The symbol's address is looked up with that device current.
static long long* offsets_scratch_for_stream(hipStream_t st, int* mismatch)
{
    int cur = -1, dev = -1;
    *mismatch = 0;
    if (hipGetDevice(&cur) != hipSuccess || cur < 0 || cur >= kMaxDevices) return nullptr;
    dev = cur;
    if (st != nullptr) {
        int sdev = -1;
        if (hipStreamGetDevice(st, &sdev) == hipSuccess && sdev >= 0 && sdev < kMaxDevices) dev = sdev;
    }
    if (dev != cur) { *mismatch = 1; return nullptr; }   // launches go to the current device: refuse
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
        zero_words(row_offsets, sizeof(int64_t), st);
        return SMLVM_OK;
    }
    if (!nnz) return SMLVM_ERR_INVALID;
    int mismatch = 0;
    long long* sums = offsets_scratch_for_stream(st, &mismatch);
    if (mismatch) return SMLVM_ERR_INVALID;   // the stream belongs to another GPU than the current one
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

extern "C" size_t smlvm_support_compact_fused_workspace_bytes(int64_t R)
{
    const int64_t ntiles = R <= 0 ? 0 : (R + kFusedTileRows - 1) / kFusedTileRows;
    return (size_t)(4 + ntiles) * sizeof(unsigned long long);
}

extern "C" int smlvm_support_compact_fused(const float* p, const int32_t* support_cols,
                                           const float* support_vals, int32_t cap,
                                           int64_t* row_offsets, int32_t* rows, int32_t* cols,
                                           float* vals, int64_t n_cap, int64_t R, int32_t K,
                                           void* workspace, size_t workspace_bytes, void* stream)
{
    if (R < 0 || K < 1 || n_cap < 0 || !row_offsets) return SMLVM_ERR_INVALID;
    if (support_cols && !(cap == 16 || cap == 32 || cap == 64)) return SMLVM_ERR_INVALID;
    if (support_cols && !aligned16(support_cols)) return SMLVM_ERR_INVALID;
    if (support_vals && (!support_cols || !aligned16(support_vals))) return SMLVM_ERR_INVALID;
    const size_t need = smlvm_support_compact_fused_workspace_bytes(R);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 7u)) return SMLVM_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    zero_words(workspace, need, st);
    if (R == 0) {
        zero_words(row_offsets, sizeof(int64_t), st);
        return SMLVM_OK;
    }
    if (!p) return SMLVM_ERR_INVALID;
    const bool vec = K % 4 == 0 && aligned16(p);
    const unsigned grid = (unsigned)((R + kFusedTileRows - 1) / kFusedTileRows);
    unsigned long long* ws = static_cast<unsigned long long*>(workspace);
#define SMLVM_FUSED(G)                                                                                      \
    if (vec) hipLaunchKernelGGL((support_compact_fused_kernel<G, true>), dim3(grid), dim3(64 * kFusedWaves), 0, st, p, support_cols, support_vals, row_offsets, rows, cols, vals, n_cap, R, K, ws); \
    else hipLaunchKernelGGL((support_compact_fused_kernel<G, false>), dim3(grid), dim3(64 * kFusedWaves), 0, st, p, support_cols, support_vals, row_offsets, rows, cols, vals, n_cap, R, K, ws)
    if (!support_cols && K <= 64) {
        hipLaunchKernelGGL((support_compact_fused_kernel<8, false, true>), dim3(grid), dim3(64 * kFusedWaves), 0, st, p, support_cols, support_vals, row_offsets, rows, cols, vals, n_cap, R, K, ws);
    }
    else if (support_cols && cap == 16) { SMLVM_FUSED(4); }
    else if (support_cols && cap == 64) { SMLVM_FUSED(16); }
    else { SMLVM_FUSED(8); }
#undef SMLVM_FUSED
    return check_launch();
}

extern "C" int smlvm_support_compact_fused_masks(const float* p, const int32_t* support_cols,
                                                 const float* support_vals, int32_t cap,
                                                 const uint32_t* support_mask, int64_t* row_offsets,
                                                 int32_t* rows, int32_t* cols, float* vals,
                                                 int64_t n_cap, int64_t R, int32_t K, void* workspace,
                                                 size_t workspace_bytes, void* stream)
{
    if (R < 0 || K < 1 || n_cap < 0 || !row_offsets) return SMLVM_ERR_INVALID;
    if (!support_cols || !support_mask || !(cap == 16 || cap == 32 || cap == 64)) return SMLVM_ERR_INVALID;
    if (!aligned16(support_cols) || !aligned16(support_mask)) return SMLVM_ERR_INVALID;
    if (support_vals && !aligned16(support_vals)) return SMLVM_ERR_INVALID;
    // the forward leaves 16-bit patterns for the 16-register tiles of its vector path only
    if (K <= 32 || K > 1024 || K % 4 != 0) return SMLVM_ERR_UNSUPPORTED;
    const size_t need = smlvm_support_compact_fused_workspace_bytes(R);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 7u)) return SMLVM_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    zero_words(workspace, need, st);
    if (R == 0) {
        zero_words(row_offsets, sizeof(int64_t), st);
        return SMLVM_OK;
    }
    if (!p) return SMLVM_ERR_INVALID;
    const bool vec = aligned16(p);
    const int lanes = K <= 128 ? 8 : (K <= 256 ? 16 : (K <= 512 ? 32 : 64));   // as smlvm_sparsemax_support_mask_bytes
    const int pat_dwords = lanes / 2;
    const unsigned grid = (unsigned)((R + kFusedTileRows - 1) / kFusedTileRows);
    unsigned long long* ws = static_cast<unsigned long long*>(workspace);
#define SMLVM_FUSED_PAT(G)                                                                                  \
    if (vec) hipLaunchKernelGGL((support_compact_fused_kernel<G, true, false, true>), dim3(grid), dim3(64 * kFusedWaves), 0, st, p, support_cols, support_vals, row_offsets, rows, cols, vals, n_cap, R, K, ws, support_mask, pat_dwords); \
    else hipLaunchKernelGGL((support_compact_fused_kernel<G, false, false, true>), dim3(grid), dim3(64 * kFusedWaves), 0, st, p, support_cols, support_vals, row_offsets, rows, cols, vals, n_cap, R, K, ws, support_mask, pat_dwords)
    if (cap == 16) { SMLVM_FUSED_PAT(4); }
    else if (cap == 64) { SMLVM_FUSED_PAT(16); }
    else { SMLVM_FUSED_PAT(8); }
#undef SMLVM_FUSED_PAT
    return check_launch();
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
        if (N <= kRedSingle) per_block = N > 0 ? N : 1;   // one block, one launch
        int nparts = (int)((N + per_block - 1) / per_block);
        if (nparts == 1) {   // small N (model-shaped batches): one launch
            hipLaunchKernelGGL(wlr_partial_kernel, dim3(1), dim3(256), 0, st, vals, comps, partials, N, per_block, inv_b, total);
        } else {
            if (nparts > 0)
                hipLaunchKernelGGL(wlr_partial_kernel, dim3((unsigned)nparts), dim3(256), 0, st, vals, comps, partials, N, per_block, 0.0f, (float*)nullptr);
            hipLaunchKernelGGL(wlr_final_kernel, dim3(1), dim3(64), 0, st, partials, nparts, inv_b, total);
        }
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

extern "C" int smlvm_support_entropy_fwd(const float* vals, int64_t N, float inv_b, float* total,
                                         void* workspace, size_t workspace_bytes, void* stream)
{
    if (N < 0 || !total) return SMLVM_ERR_INVALID;
    if (N > 0 && !vals) return SMLVM_ERR_INVALID;
    if (!workspace || workspace_bytes < smlvm_weighted_loss_reduce_workspace_bytes(N)) return SMLVM_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double* partials = (double*)workspace;
    int64_t per_block = (N + kRedBlocks - 1) / kRedBlocks;
    if (per_block < 256) per_block = 256;
    if (N <= kRedSingle) per_block = N > 0 ? N : 1;   // one block, one launch
    const int nparts = (int)((N + per_block - 1) / per_block);
    if (nparts == 1) {
        hipLaunchKernelGGL(plogp_partial_kernel, dim3(1), dim3(256), 0, st, vals, partials, N, per_block, -inv_b, total);
    } else {
        if (nparts > 0)
            hipLaunchKernelGGL(plogp_partial_kernel, dim3((unsigned)nparts), dim3(256), 0, st, vals, partials, N, per_block, 0.0f, (float*)nullptr);
        hipLaunchKernelGGL(wlr_final_kernel, dim3(1), dim3(64), 0, st, partials, nparts, -inv_b, total);
    }
    return check_launch();
}

extern "C" int smlvm_support_entropy_bwd(const float* vals, const float* gtotal, float inv_b,
                                         float* dvals, int64_t N, void* stream)
{
    if (N < 0) return SMLVM_ERR_INVALID;
    if (N == 0) return SMLVM_OK;
    if (!vals || !gtotal || !dvals) return SMLVM_ERR_INVALID;
    int64_t blocks = (N + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(plogp_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, vals, gtotal, inv_b, dvals, N);
    return check_launch();
}

extern "C" int smlvm_support_scatter(const float* g, const int64_t* row_offsets, const int32_t* cols,
                                     float* out, int64_t R, int32_t K, void* stream)
{
    if (R < 0 || K < 1) return SMLVM_ERR_INVALID;
    if (R == 0) return SMLVM_OK;
    if (!row_offsets || !out) return SMLVM_ERR_INVALID;   // (g / cols may be NULL only if N == 0: not checkable here)
    if (!g || !cols) return SMLVM_ERR_INVALID;
    if ((R + 3) / 4 > 0x7fffffffll) return SMLVM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(support_scatter_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, row_offsets, cols, out, R, K);
    return check_launch();
}

extern "C" int smlvm_support_segment_sum(const float* g, const int64_t* row_offsets, float* out,
                                         int64_t R, int64_t D, void* stream)
{
    if (R < 0 || D < 1) return SMLVM_ERR_INVALID;
    if (R == 0) return SMLVM_OK;
    if (!row_offsets || !out || !g) return SMLVM_ERR_INVALID;
    if (R > 65535) return SMLVM_ERR_UNSUPPORTED;          // (grid.y; batches, not matrices of rows)
    const bool vec = D % 4 == 0 && aligned16(g) && aligned16(out);
    const int64_t cols = vec ? D / 4 : D;
    if ((cols + 255) / 256 > 0x7fffffffll) return SMLVM_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((cols + 255) / 256), (unsigned)R);
    if (vec) hipLaunchKernelGGL(support_segment_sum_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, g, row_offsets, out, R, D);
    else hipLaunchKernelGGL(support_segment_sum_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, g, row_offsets, out, R, D);
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
