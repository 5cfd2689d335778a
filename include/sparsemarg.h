/*
 * sparsemarg.h -- C ABI of the MI355X-native sparse-marginalization hot path.
 *
 * Drop-in boundary for the hot path of deep-spin/sparse-marginalization-lvm (`lvmhelpers`):
 * every entry point takes raw DEVICE pointers, sizes and a HIP stream (`void* stream` is a
 * `hipStream_t`; NULL = the default stream), launches on that stream, never synchronises, never
 * allocates (scratch comes in through explicit `workspace` arguments whose size the matching
 * `*_workspace_bytes` function reports) and returns 0 or a negative SMLVM_ERR_* code.  No torch
 * types appear here; the Python host side (`lvmhelpers/_native.py`) binds these with ctypes.
 * Citations are file:line into the reference repository.
 *
 * Library: sparse-marginalization-lvm_amd/csrc/libsparsemarg_hip.so (gfx950 code object).
 */
#ifndef SPARSEMARG_H_
#define SPARSEMARG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMLVM_ABI_VERSION 2

#define SMLVM_OK 0
#define SMLVM_ERR_INVALID (-1)     /* null pointer, negative size, bad combination of arguments */
#define SMLVM_ERR_UNSUPPORTED (-2) /* size outside what the kernels are built for */
#define SMLVM_ERR_HIP (-3)         /* the HIP runtime refused the launch (see smlvm_last_hip_error) */
#define SMLVM_ERR_WORKSPACE (-4)   /* workspace missing or too small */

/* Limits (smlvm_* returns SMLVM_ERR_UNSUPPORTED beyond them). */
#define SMLVM_TOPK_MAX_BITS 512 /* lvmhelpers/binary_topk.h:13 CODE_MAX_SIZE */
#define SMLVM_TOPK_MAX_K 1024 /* k <= 64: a lane group per sample; beyond: a workgroup per sample */
#define SMLVM_SPARSEMAP_MAX_BITS 128

int smlvm_abi_version(void);
const char* smlvm_error_string(int code);
/* hipError_t (as int) of the last failed launch on the calling thread, 0 if none. */
int smlvm_last_hip_error(void);

/* ---------------------------------------------------------------------------------------------
 * Kernel 1: batched row-wise sparsemax.
 * Replaces entmax.sparsemax(X, dim=-1) as called at lvmhelpers/marg.py:45,51 and
 * lvmhelpers/structmarg.py:48 (forward) and its autograd backward.
 *   x, p, dy, dx : [R, K] fp32 row-major contiguous;  supp_size : [R] int32 (entmax's support_size,
 *   the divisor of the backward);  optional outputs (NULL to skip): nnz [R] int32 = #{p > 0},
 *   tau [R] fp32 (threshold in the max-shifted row), entropy [R] fp32 (lvmhelpers/marg.py:6-28).
 * p = max(x - max(x) - tau, 0).  dx = [p != 0] * (dy - sum(dy[p != 0]) / supp_size).
 * Inputs and outputs must not overlap.  Any K >= 1: rows of up to 65536 elements are held in
 * registers (one read, one write); wider rows are streamed (three reads, one write).
 */
int smlvm_sparsemax_fwd(const float* x, float* p, int32_t* supp_size, int32_t* nnz, float* tau,
                        float* entropy, int64_t R, int32_t K, void* stream);
int smlvm_sparsemax_bwd(const float* dy, const float* p, const int32_t* supp_size, float* dx,
                        int64_t R, int32_t K, void* stream);

/* The same pair with a by-product that lets the backward skip p: the forward also writes the
 * columns of the non-zeros of every row, the backward gathers dy at those columns and writes dx once
 * (no read of p, no streaming read of dy: the backward of sparsemax is O(support) reads + one dense
 * write).  What entmax keeps on the autograd ctx is (supp_size, p); this keeps (supp_size, lists).
 *   support_cols [R, cap] int32, 1 <= cap <= 64 (<= 128 for K > 4096): columns with p > 0 in no particular order, -1 in
 *   unused slots; first entry -2 = no list for this row (more than cap non-zeros, or a row whose
 *   threshold search could not run on a candidate list: more than 256 elements within 1 of the row
 *   maximum after the bound has been raised; for K > 4096 more than 64 of them in one wave's 64 E
 *   columns): the backward then reads p for that row exactly like smlvm_sparsemax_bwd.
 *   support_cols may be NULL (then these are the plain entry points).
 *   support_mask (nullable; smlvm_sparsemax_support_mask_bytes(R, K) bytes, 0 unless
 *   32 < K <= 4096 -- a row of at most 32 entries always fits its list): non-zero patterns as
 *   bitmaps (layout private to the library).  1024 < K <= 4096: rows with more than cap non-zeros
 *   leave their pattern (512 B) and are marked -3 instead of -2; the backward gathers dy at those
 *   bits and does not read p (rows with up to 96 non-zeros).  32 < K <= 1024: EVERY row leaves its
 *   pattern (2 bytes per lane of its lane group: 16 B ... 128 B per row), rows with more than cap
 *   non-zeros are marked -3, and the backward streams dy for them without reading p.  Pass the
 *   same buffer, with the same R and K, to both calls, or NULL to both. */
size_t smlvm_sparsemax_support_mask_bytes(int64_t R, int32_t K);
int smlvm_sparsemax_fwd_cols(const float* x, float* p, int32_t* supp_size, int32_t* nnz, float* tau,
                             float* entropy, int32_t* support_cols, int32_t cap,
                             uint32_t* support_mask, int64_t R, int32_t K, void* stream);
int smlvm_sparsemax_bwd_cols(const float* dy, const int32_t* support_cols, int32_t cap,
                             const uint32_t* support_mask, const float* p,
                             const int32_t* supp_size, float* dx, int64_t R, int32_t K,
                             void* stream);
/* smlvm_sparsemax_fwd_cols that also leaves the VALUES of the listed non-zeros:
 *   support_vals [R, cap] fp32, slot for slot with support_cols (slots without a column are not
 *   written); the same bits as p at those columns.  With them smlvm_support_compact_fused gathers
 *   nothing from p: this is the sparsemax epilogue that feeds the compaction of
 *   lvmhelpers/structmarg.py:113-126 / the support-only mode of marg.py:86-105.
 * K <= 4096 and cap <= 64 (the one-wave-per-row kernels; SMLVM_ERR_UNSUPPORTED beyond);
 * support_vals == NULL: exactly smlvm_sparsemax_fwd_cols. */
int smlvm_sparsemax_fwd_lists(const float* x, float* p, int32_t* supp_size, int32_t* nnz, float* tau,
                              float* entropy, int32_t* support_cols, float* support_vals,
                              int32_t cap, uint32_t* support_mask, int64_t R, int32_t K,
                              void* stream);

/* Shannon entropy with exact zeros, lvmhelpers/marg.py:6-28 (also structmarg.py:9-20):
 * ent[r] = -sum_{p>0} pc*log(pc), pc = clamp(p, eps, 1-eps), eps = FLT_EPSILON. */
int smlvm_entropy(const float* p, float* ent, int64_t R, int32_t K, void* stream);

/* 1.5-entmax over the last axis: third-party entmax.entmax15, the default normaliser of
 * ExplicitWrapper (lvmhelpers/marg.py:34,46) and the mapping inside Entmax15Loss
 * (experiments/semi_supervised-vae/train.py:9,314-315).
 *   p = max(z - tau, 0)^2,  z = x/2 - max(x/2),  tau: sum p = 1
 *   x [R,K] fp32 -> p [R,K] fp32; nnz [R] int32 = #{p > 0} and tau [R] fp32 (frame of z) nullable.
 *   bwd: dx = sqrt(p) (dy - sum dy sqrt(p) / sum sqrt(p))       (Entmax15Function.backward)
 * Floating-point parity with the package's sort/cumsum algorithm (2e-6 on p), not bit parity.
 * K <= 65536 (SMLVM_ERR_UNSUPPORTED beyond). */
int smlvm_entmax15_fwd(const float* x, float* p, int32_t* nnz, float* tau, int64_t R, int32_t K,
                       void* stream);
/* The same pair with support lists, for rows held by one wave (512 < K <= 4096; SMLVM_ERR_UNSUPPORTED
 * otherwise): the forward also writes support_cols [R, cap] int32 (columns with p > 0, -1 padding,
 * -2 in slot 0 = no list: more than cap non-zeros, or a row whose threshold search ran full width)
 * and support_vals [R, cap] fp32 = sqrt(p) of the listed columns, slot for slot; 1 <= cap <= 64.  The
 * backward gathers dy at the listed columns, writes dx = sqrt(p) (dy - q) there and zeros elsewhere,
 * and reads p only for rows without a list (same Entmax15Function.backward, O(support) reads). */
int smlvm_entmax15_fwd_lists(const float* x, float* p, int32_t* nnz, float* tau, int32_t* support_cols,
                             float* support_vals, int32_t cap, int64_t R, int32_t K, void* stream);
int smlvm_entmax15_bwd_lists(const float* dy, const float* p, const int32_t* support_cols,
                             const float* support_vals, int32_t cap, float* dx, int64_t R, int32_t K,
                             void* stream);
int smlvm_entmax15_bwd(const float* dy, const float* p, float* dx, int64_t R, int32_t K,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * Kernel 2: non-zero-support stream compaction + probability-weighted loss reduction.
 * Replaces the boolean-mask compaction of lvmhelpers/structmarg.py:113-126 (mask = p.view(-1) > 0,
 * row-major order identical to torch.nonzero) and the `p @ loss` contractions of
 * lvmhelpers/marg.py:125, structmarg.py:140,242.
 *   support_count   : nnz[r] = #{p[r,:] > 0}
 *   support_offsets : row_offsets[0..R] = exclusive prefix sum of nnz (row_offsets[R] = N), int64
 *   support_compact : for the j-th positive entry of row r (ascending column):
 *                     rows[row_offsets[r]+j] = r, cols[..] = column, vals[..] = p[r, column]
 *                     (any of rows/cols/vals may be NULL)
 *   weighted_loss_reduce_fwd : row_out[r] = sum_{j in row r} vals[j]*comps[j] (nullable),
 *                     total[0] = inv_b * sum_j vals[j]*comps[j]  (deterministic order, fp64 accumulate)
 *   weighted_loss_reduce_bwd : dvals[j] = g*inv_b*comps[j], dcomps[j] = g*inv_b*vals[j], g = gtotal[0]
 *                     (either output may be NULL)
 */
int smlvm_support_count(const float* p, int32_t* nnz, int64_t R, int32_t K, void* stream);
size_t smlvm_support_offsets_workspace_bytes(int64_t R);
int smlvm_support_offsets(const int32_t* nnz, int64_t* row_offsets, int64_t R, void* workspace,
                          size_t workspace_bytes, void* stream);
int smlvm_support_compact(const float* p, const int64_t* row_offsets, int32_t* rows, int32_t* cols,
                          float* vals, int64_t R, int32_t K, void* stream);
/* smlvm_support_offsets without a workspace argument.  No counterpart in the reference: it exists
 * for callers that cannot allocate device memory -- code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  The block sums of the scan live in a static device
 * array of the library (64 KiB per GPU, held until the process ends), so R is limited to
 * SMLVM_NOALLOC_MAX_ROWS rows (SMLVM_ERR_UNSUPPORTED beyond).  Same result as
 * smlvm_support_offsets, bit for bit.  Nothing is allocated or freed inside (the first call on a
 * device looks the array's address up once, with that device current).  Every call on one GPU
 * uses that GPU's one array: calls that overlap on the same GPU (two streams) race -- serialise
 * them, or use smlvm_support_offsets with a workspace per stream.  The stream's GPU must be the
 * current device of the calling thread (as for every launch here); a stream of another GPU is
 * refused with SMLVM_ERR_INVALID rather than handed the current GPU's array. */
#define SMLVM_NOALLOC_MAX_ROWS (1 << 24)
int smlvm_support_offsets_noalloc(const int32_t* nnz, int64_t* row_offsets, int64_t R, void* stream);
/* The same two steps from the forward's support lists (smlvm_sparsemax_fwd_cols): rows that have
 * a list are counted / compacted from its <= cap entries (sorted by column on the fly, so the
 * output order is unchanged) and only gather their probabilities; rows marked -2 are scanned.
 * This is the "fused into the sparsemax epilogue" form of the reference's boolean-mask compaction
 * (lvmhelpers/structmarg.py:113-126): 12 N + 4 cap R bytes instead of 4 K R + 12 N.
 * These two take cap <= 64 (SMLVM_ERR_INVALID beyond): the 128-slot lists that
 * smlvm_sparsemax_fwd_cols can write for K > 4096 feed the backward only -- count / compact such
 * rows from p (support_cols = NULL). */
int smlvm_support_count_cols(const float* p, const int32_t* support_cols, int32_t cap, int32_t* nnz,
                             int64_t R, int32_t K, void* stream);
int smlvm_support_compact_cols(const float* p, const int32_t* support_cols, int32_t cap,
                               const int64_t* row_offsets, int32_t* rows, int32_t* cols, float* vals,
                               int64_t R, int32_t K, void* stream);
/* Count + offsets + compaction in ONE launch (round 3): the boolean-mask compaction of
 * lvmhelpers/structmarg.py:113-126 (and marg.py:86-87's live-column test) as a single pass over the
 * forward's lists.  Same row_offsets / rows / cols / vals as the three calls above, bit for bit.
 *   support_cols [R, cap] nullable (cap in {16, 32, 64}, 16-byte aligned): rows with a list are
 *       taken from it, rows marked -2 / -3 and all rows when it is NULL are scanned from p;
 *   support_vals [R, cap] nullable (smlvm_sparsemax_fwd_lists): values of the listed entries; NULL:
 *       they are gathered from p;
 *   row_offsets [R+1] int64: always complete, row_offsets[R] = N;
 *   rows / cols / vals (each nullable): capacity n_cap entries each -- entries at positions >=
 *       n_cap are dropped, so a caller that does not know N sizes them by an upper bound (R * cap
 *       when every row has a list) or calls again with n_cap = N and the offsets it now has;
 *   workspace: smlvm_support_compact_fused_workspace_bytes(R) bytes, 8-byte aligned, cleared by the
 *       call itself; afterwards word 2 (uint64) holds N and word 1 is non-zero if a tile gave up
 *       waiting for the tiles before it (bounded wait; every result of that launch is void then).
 * Tiles of 512 rows exchange their totals through the workspace (decoupled look-back): no second
 * pass over the lists, no host round trip between counting and compacting. */
size_t smlvm_support_compact_fused_workspace_bytes(int64_t R);
int smlvm_support_compact_fused(const float* p, const int32_t* support_cols, const float* support_vals,
                                int32_t cap, int64_t* row_offsets, int32_t* rows, int32_t* cols,
                                float* vals, int64_t n_cap, int64_t R, int32_t K, void* workspace,
                                size_t workspace_bytes, void* stream);
/* smlvm_support_compact_fused with the forward's patterns: for 32 < K <= 1024 (K % 4 == 0)
 * smlvm_sparsemax_fwd_cols / _fwd_lists leave, in `support_mask`, 16 bits per lane of EVERY row's
 * non-zero pattern (smlvm_sparsemax_support_mask_bytes(R, K) bytes) when x and p are 16-byte aligned.
 * Rows without a list (more non-zeros than cap) are then counted from those 2-64 bytes instead of
 * from their K probabilities; the entries are still read from p.  Same outputs.  support_cols and
 * support_mask must come from the same forward call as p (a mask buffer that the forward did not
 * fill -- unaligned x, K % 4 != 0 -- gives wrong counts: the caller's to know). */
int smlvm_support_compact_fused_masks(const float* p, const int32_t* support_cols,
                                      const float* support_vals, int32_t cap,
                                      const uint32_t* support_mask, int64_t* row_offsets,
                                      int32_t* rows, int32_t* cols, float* vals, int64_t n_cap,
                                      int64_t R, int32_t K, void* workspace, size_t workspace_bytes,
                                      void* stream);
/* Entropy of the compacted distribution and the gather it is taken over.  Replaces
 * lvmhelpers/structmarg.py:51-54 and :200 (`-(distr @ torch.log(distr)) / batch_size`, seven
 * elementwise launches forward and six backward) and the advanced indexing `p[rows, cols]` of
 * structmarg.py:123-126,189-190 whose backward sorts its indices:
 *   entropy_fwd: total[0] = -inv_b * sum_{vals[j] > 0} vals[j] log vals[j] (fp64 two-stage sum, fixed
 *                order; workspace: smlvm_weighted_loss_reduce_workspace_bytes(N) bytes);
 *   entropy_bwd: dvals[j] = -(log vals[j] + 1) * inv_b * gtotal[0] for vals[j] > 0, else 0;
 *   scatter:     out [R, K] = 0 except out[r, cols[j]] = g[j] for j in [row_offsets[r],
 *                row_offsets[r+1]) -- the backward of vals = p[rows, cols] (positions are distinct). */
int smlvm_support_entropy_fwd(const float* vals, int64_t N, float inv_b, float* total, void* workspace,
                              size_t workspace_bytes, void* stream);
int smlvm_support_entropy_bwd(const float* vals, const float* gtotal, float inv_b, float* dvals,
                              int64_t N, void* stream);
int smlvm_support_scatter(const float* g, const int64_t* row_offsets, const int32_t* cols, float* out,
                          int64_t R, int32_t K, void* stream);
/* out [R, D] = per-row sums of g [N, D] over each row's compacted entries [row_offsets[r],
 * row_offsets[r+1]), in entry order (fixed-order fp32 sums; entries at or beyond row_offsets[R] are
 * ignored): the backward of gathering one vector per compacted entry from a per-row table, e.g. the
 * signal game's candidate embeddings `embeddings[rows]` (experiments/signal-game/archs.py:76-89 run
 * on the support's (row, symbol) pairs), which torch's index_select backward sums with float
 * atomics.  R <= 65535 (SMLVM_ERR_UNSUPPORTED beyond). */
int smlvm_support_segment_sum(const float* g, const int64_t* row_offsets, float* out, int64_t R,
                              int64_t D, void* stream);
size_t smlvm_weighted_loss_reduce_workspace_bytes(int64_t N);
int smlvm_weighted_loss_reduce_fwd(const float* vals, const float* comps, const int64_t* row_offsets,
                                   float* row_out, float* total, float inv_b, int64_t R, int64_t N,
                                   void* workspace, size_t workspace_bytes, void* stream);
int smlvm_weighted_loss_reduce_bwd(const float* vals, const float* comps, const float* gtotal,
                                   float inv_b, float* dvals, float* dcomps, int64_t N,
                                   void* stream);

/* ---------------------------------------------------------------------------------------------
 * k-best bit vectors. Replaces lvmhelpers/pbinary_topk.pyx:23-34 batched_topk (which calls
 * lvmhelpers/binary_topk.h:100-135 per row on the host).
 *   scores [B, n] fp32 -> configs [B, k, n] fp32 in {0,1}, sorted by descending score
 *   (ties: the reference's merge order, binary_topk.h:72); topscores [B, k] fp32 (nullable) are
 *   the fp32 sums accumulated in index order (binary_topk.h:28).
 * Requires 2 <= k <= min(2^n, SMLVM_TOPK_MAX_K), 1 <= n <= SMLVM_TOPK_MAX_BITS (the reference has
 * undefined behaviour outside 2 <= k <= 2^n; this returns SMLVM_ERR_INVALID instead).
 */
int smlvm_binary_topk_batched(const float* scores, float* configs, float* topscores, int64_t B,
                              int32_t n, int32_t k, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Kernel 3: bit-vector SparseMAP, one wavefront per sample.
 * Replaces, for a whole batch at once, lvmhelpers/sparsemap.py:12-36 (run_sparsemap: factor graph
 * construction, optional init_active_set_from_scores, lpsmap GenericFactor::SolveQP,
 * get_sparse_solution) with the FactorBernoulli / FactorBudget oracles of
 * lvmhelpers/bernoulli.h:14-109 and lvmhelpers/budget.h:24-85, and lvmhelpers/sparsemap.py:38-44
 * (jv -> dist_jacobian_vec) for the backward.
 *   scores      [B, n] fp32            (eta = [scores.double(); 0], sparsemap.py:21-22)
 *   init_scores [B, 2n] fp64 or NULL   (torch.rand(2n, dtype=double) per sample, sparsemap.py:25-27)
 *   budget      0 = Bernoulli factor, >0 = budget factor;  max_iter as in solve_qp(max_iter=)
 * With cap = n+1, W = ceil(n/32), ld = n+2:
 *   aset_bits [B, cap, W] uint32   active configurations (bit k%32 of word k/32 = variable k on)
 *   p         [B, cap] fp32        distribution over the active set, zero beyond alen[b]
 *   alen      [B] int32            len(aset)
 *   inv_a     [B, ld, ld] fp64     cached inverse KKT matrix (rows/cols 0..alen[b] written); the
 *                                  state the reference keeps on ctx.f between forward and backward
 *   iters     [B] int32, status [B] int32 (0 converged, 1 max_iter, 2 singular, 3 duplicate,
 *                                  4 stalled: no accurate KKT solution within the polishing budget,
 *                                  see oracle/sparsemap_oracle.c C7); nullable.  A converged sample has
 *                                  sum p = 1 and the optimal marginals to the precision of the fp32 output
 *   workspace : smlvm_sparsemap_fwd_workspace_bytes(B, n, inv_a != NULL) bytes, 8-byte aligned
 *               (0 for n <= 64: may be NULL).  For n > 64 the batch runs in tiers of growing LDS
 *               footprint (4, 2, 1 samples per CU); a sample whose active set outgrows a tier parks
 *               its solver state here and the next launch resumes it -- results do not depend on it.
 * Backward: dp [B, cap] fp32 -> dscores [B, n] fp32.
 */
size_t smlvm_sparsemap_fwd_workspace_bytes(int64_t B, int32_t n, int32_t with_inv_a);
int smlvm_sparsemap_fwd(const float* scores, const double* init_scores, int32_t budget,
                        int32_t max_iter, int64_t B, int32_t n, uint32_t* aset_bits, float* p,
                        int32_t* alen, double* inv_a, int32_t* iters, int32_t* status,
                        void* workspace, size_t workspace_bytes, void* stream);
int smlvm_sparsemap_bwd(const float* dp, const uint32_t* aset_bits, const int32_t* alen,
                        const double* inv_a, int64_t B, int32_t n, float* dscores, void* stream);

/* The binary sequence factor on the same solver (sequence_smap, lvmhelpers/sparsemap.py:47-87,
 * 130-148,159-160; FactorSequenceBinary of lvmhelpers/sequence_binary.h:11-67 on sequence.h:66-166):
 * n positions with states {0,1}; trans [B, n-1] fp32 are the additional log-potentials
 * (set_additional_log_potentials, sparsemap.py:62-63), the score of the (1 -> 1) transition into
 * position k+1.  Linear oracle = Viterbi (first state wins ties), Evaluate adds the edge scores,
 * InitActiveSetFromScores runs on init_scores with zero edge potentials (sparsemap.py:67-69).  Edge
 * potentials are not functions of the variable marginals, so active sets can be affinely dependent
 * in variable space: a singular insertion swaps a vertex out (oracle/sparsemap_oracle.c C8) where
 * the bit-vector factors report status 2.  2 <= n <= SMLVM_SPARSEMAP_MAX_BITS; outputs, workspace
 * and status codes as smlvm_sparsemap_fwd.  Backward (SequenceSparseMAP.jv, sparsemap.py:79-87):
 * dp [B, cap] -> dscores [B, n] and dtrans [B, n-1]. */
int smlvm_sparsemap_seq_fwd(const float* scores, const float* trans, const double* init_scores,
                            int32_t max_iter, int64_t B, int32_t n, uint32_t* aset_bits, float* p,
                            int32_t* alen, double* inv_a, int32_t* iters, int32_t* status,
                            void* workspace, size_t workspace_bytes, void* stream);
int smlvm_sparsemap_seq_bwd(const float* dp, const uint32_t* aset_bits, const int32_t* alen,
                            const double* inv_a, int64_t B, int32_t n, float* dscores,
                            float* dtrans, void* stream);

/* Expand compacted (sample, slot) pairs into 0/1 float rows: out[j, :] = bits of
 * aset_bits[rows[j], cols[j], :]  (the `samplei[distri > 0]` gather + torch.cat of
 * lvmhelpers/structmarg.py:189,198).  out [N, n] fp32. */
int smlvm_unpack_configs(const uint32_t* aset_bits, const int32_t* rows, const int32_t* cols,
                         float* out, int64_t N, int32_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Next rows after the hot path: the callers on the decoder side.
 *
 * Low-rank categorical negative log-likelihood without materialised logits.  Replaces, as a pair,
 * CategoricalGenerator.forward's `self.out(X)` (experiments/bit_vector-vae/archs.py:26-29) and
 * reconstruction_loss (experiments/bit_vector-vae/train.py:361-371):
 *   logits[n,f,c] = b[c] + sum_r x[n,f,r] w[c,r];   loss[n] = sum_f (logsumexp_c logits - logits[n,f,y[n,f]])
 *   x [N,F,R] fp32, w [C,R] fp32 (nn.Linear weight), b [C] fp32, y [N,F] labels in [0,C)
 *   y_kind: 0 = uint8, 1 = int32, 2 = int64.   Limits: R <= 8, C <= 256.
 *   A label outside [0, C) never indexes w or b: the forward writes NaN into that pair's nll (hence
 *   the row's loss), the backward NaN into its dx; the reference's F.cross_entropy raises a device
 *   assert in that case (experiments/bit_vector-vae/train.py:368).
 *   fwd outputs: lse [N,F] fp32 (saved for the backward), nll [N,F] fp32, loss [N] fp32 (nullable)
 *   bwd: gloss [N] = dL/dloss;  dx [N,F,R], dw [C,R], db [C] (each nullable); workspace of
 *        smlvm_lowrank_ce_bwd_workspace_bytes bytes for the per-block dw/db partials.
 *   fwd_dxdir: the forward with one more output, dxdir [N,F,R] fp32 =
 *        sum_c softmax(logits[n,f,:])_c w[c,:] - w[y[n,f],:], the gradient of loss[n] with respect to
 *        x[n,f,:]: the training backward's dx is gloss[n] * dxdir[n,f,:] and `bwd` is then called
 *        with dx = NULL (it costs the forward R packed FMAs more per logit and saves the backward a
 *        whole pass of C exponentials per element).
 */
int smlvm_lowrank_ce_fwd_dxdir(const float* x, const float* w, const float* b, const void* y,
                               int32_t y_kind, float* lse, float* nll, float* loss, float* dxdir,
                               int64_t N, int32_t F, int32_t C, int32_t R, void* stream);
int smlvm_lowrank_ce_fwd(const float* x, const float* w, const float* b, const void* y,
                         int32_t y_kind, float* lse, float* nll, float* loss, int64_t N, int32_t F,
                         int32_t C, int32_t R, void* stream);
size_t smlvm_lowrank_ce_bwd_workspace_bytes(int64_t N, int32_t F, int32_t C, int32_t R);
int smlvm_lowrank_ce_bwd(const float* x, const float* w, const float* b, const void* y,
                         int32_t y_kind, const float* lse, const float* gloss, float* dx, float* dw,
                         float* db, int64_t N, int32_t F, int32_t C, int32_t R, void* workspace,
                         size_t workspace_bytes, void* stream);

/* Segmented log-sum-exp over the compacted support lists.  Replaces the per-sample Python loop of
 * compute_importance_sampling (experiments/bit_vector-vae/train.py:325-330:
 * `torch.logsumexp(logp_x_given_z[torch.tensor(idxs) == k] + logp_z, dim=0)` for every k):
 *   out[r] = log sum_{j in [row_offsets[r], row_offsets[r+1])} exp(v[j] + shift);  -inf for an
 *   empty segment.  v [N] fp32, row_offsets [R+1] int64 (as produced by smlvm_support_offsets). */
int smlvm_segmented_logsumexp(const float* v, const int64_t* row_offsets, float shift, float* out,
                              int64_t R, void* stream);

/* ---- HIP-graph repair pass (no counterpart in the reference; for lvmhelpers/graphs.py) ------------
 * Replaces every memset node of a captured hipGraph_t (child graphs included) by a kernel node that
 * writes the same bytes, with the same dependencies.  With the HIP runtime inside this image's
 * PyTorch wheel (7.0.51831) a captured memset writes the captured value on the first launch of the
 * instantiated graph only and an arbitrary word afterwards; PyTorch's own reductions (semaphores zeroed with cudaMemsetAsync) then return wrong
 * sums from the second replay on.  Call it on the graph BEFORE instantiating it.
 * n_memsets / n_replaced (may be NULL): memset nodes found / replaced.  SMLVM_ERR_UNSUPPORTED for
 * an element size other than 1, 2, 4; SMLVM_ERR_HIP when a graph API call fails (the graph may
 * then be partly rewritten: discard it). */
int smlvm_graph_memsets_to_kernels(void* graph /* hipGraph_t */, int32_t* n_memsets, int32_t* n_replaced);

#ifdef __cplusplus
}
#endif
#endif /* SPARSEMARG_H_ */
