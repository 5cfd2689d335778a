/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the k-best bit-vector enumerator:
 *     lvmhelpers/binary_topk.h:18-46   scored_cfg {float score; bitset<512> cfg}, update()
 *     lvmhelpers/binary_topk.h:57-98   merge_branch (list merge, b taken only if STRICTLY greater)
 *     lvmhelpers/binary_topk.h:100-129 topk (DP over variables, k-bounded lists)
 *     lvmhelpers/pbinary_topk.pyx:23-34 batched_topk (fills caller-allocated [B,k,n] float 0/1)
 * Scores are fp32 sums accumulated in index order (binary_topk.h:16,28).
 *
 * Pinned against the compiled reference (oracle/_ref, built from the reference's own header) by
 * tests/test_oracle_binary_topk.py and the golden fixtures under tests/golden/.
 *
 * Differences from the reference, on purpose: k > 2^n (reference: reads past the vector, UB,
 * pbinary_topk.pyx:33-34) and k < 2 (reference: UB, binary_topk.h:102 assert compiled out)
 * return an error instead.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_TOPK_MAX_BITS 512
#define ORACLE_TOPK_WORDS (ORACLE_TOPK_MAX_BITS / 32)

typedef struct {
    float score;
    uint32_t cfg[ORACLE_TOPK_WORDS];
} scored_t;

static void set_on(scored_t* out, const scored_t* in, unsigned i, float v)
{
    scored_t t = *in;
    t.score = in->score + v; /* binary_topk.h:28 */
    t.cfg[i >> 5] |= (1u << (i & 31));
    *out = t;
}

/* Runs the DP for one row. `a` and `b` are scratch lists of k entries; returns a pointer to the
 * one holding the result and its length in *len_out. */
static const scored_t* topk_row(const float* x, int n, int k, scored_t* a, scored_t* b, int* len_out)
{
    scored_t c0, c1, tmp;
    scored_t *curr = a, *next = b;
    memset(&c0, 0, sizeof c0);
    set_on(&c1, &c0, 0, x[0]);
    if (x[0] >= 0) { curr[0] = c1; curr[1] = c0; } /* binary_topk.h:110-116 */
    else           { curr[0] = c0; curr[1] = c1; }
    int len = 2;
    for (int i = 1; i < n; ++i) {
        int ia = 0, ib = 0, ins = 0;
        while (ins < k && ia < len && ib < len) { /* binary_topk.h:70-81 */
            set_on(&tmp, &curr[ib], (unsigned)i, x[i]);
            if (tmp.score > curr[ia].score) { next[ins] = tmp; ++ib; }
            else                            { next[ins] = curr[ia]; ++ia; }
            ++ins;
        }
        while (ins < k && ia < len) next[ins++] = curr[ia++];                           /* :83-88 */
        while (ins < k && ib < len) set_on(&next[ins++], &curr[ib++], (unsigned)i, x[i]); /* :90-95 */
        scored_t* t = curr; curr = next; next = t;                                      /* :125-126 */
        len = ins;
    }
    *len_out = len;
    return curr;
}

/* scores[B,n] -> configs[B,k,n] (0/1 floats), topscores[B,k] (nullable). */
int oracle_binary_topk_batched(const float* scores, float* configs, float* topscores,
                               int64_t B, int32_t n, int32_t k)
{
    if (k < 2 || n < 1 || n > ORACLE_TOPK_MAX_BITS) return -2;
    if (n < 31 && (int64_t)k > ((int64_t)1 << n)) return -3;
    scored_t* buf0 = (scored_t*)malloc(sizeof(scored_t) * (size_t)k);
    scored_t* buf1 = (scored_t*)malloc(sizeof(scored_t) * (size_t)k);
    if (!buf0 || !buf1) { free(buf0); free(buf1); return -1; }
    for (int64_t r = 0; r < B; ++r) {
        int len = 0;
        const scored_t* fin = topk_row(scores + r * n, n, k, buf0, buf1, &len);
        for (int j = 0; j < k; ++j) {
            for (int i = 0; i < n; ++i)
                configs[(r * k + j) * (int64_t)n + i] =
                    (j < len) ? (float)((fin[j].cfg[i >> 5] >> (i & 31)) & 1u) : 0.0f;
            if (topscores) topscores[r * k + j] = (j < len) ? fin[j].score : 0.0f;
        }
    }
    free(buf0); free(buf1);
    return 0;
}
