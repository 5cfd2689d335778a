// ORACLE BUILD GLUE (test infrastructure, NOT product code).
//
// Thin C wrapper that compiles the REFERENCE's own header where it lies
// (/root/reference/lvmhelpers/binary_topk.h, found through -I, never copied) into
// oracle/_ref/libref_binary_topk.so. The loop below does what the reference's Cython entry point
// does (lvmhelpers/pbinary_topk.pyx:23-34: topk(&scores[i,0], size, k) then populate_out per
// configuration); it also returns the scores the reference computes (scored_cfg::score).
// Only built when /root/reference is present (this container); the GPU box uses the prebuilt .so.
#include "binary_topk.h"

#include <cstdint>

extern "C" int ref_binary_topk_batched(const float* scores, float* configs, float* topscores,
                                       int64_t B, int32_t n, int32_t k)
{
    for (int64_t i = 0; i < B; ++i) {
        std::vector<scored_cfg> out = topk(const_cast<float*>(scores + i * n), n, k);
        for (int32_t j = 0; j < k && j < (int32_t)out.size(); ++j) {
            out[j].populate_out(configs + (i * k + j) * (int64_t)n, n);
            if (topscores) topscores[i * k + j] = out[j].score;
        }
    }
    return 0;
}
