/* Per-pass trace of the SparseMAP oracle (diagnosis of slow / non-converging samples; round 3).
 *
 *   gcc -O2 -ffp-contract=off -o /tmp/sm_trace tools/sm_trace.c -lm
 *   /tmp/sm_trace sample.bin 2>&1 | less
 *
 * sample.bin: int32 n, int32 budget, int32 max_iter, n floats (scores), 2n doubles (init scores);
 * budget -1 = the binary sequence factor, n-1 floats (edge potentials) follow the init scores;
 * tools/sm_trace_dump.py writes one per sample that ends with a status other than CONVERGED.
 * Builds oracle/sparsemap_oracle.c with -DSM_TRACE: every pass prints what it did (ORACLE value - tau,
 * INSERT with the Schur complement s, REMOVE with alpha, FULLSTEP, POLISH / C7 residuals, C10).
 * Test infrastructure like the oracle itself: nothing in the product builds or loads this. */
#define SM_TRACE 1
#include "../oracle/sparsemap_oracle.c"
#include <stdio.h>

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s sample.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int n, budget, mi;
    float x[SM_MAX_N];
    double init[2 * SM_MAX_N];
    if (fread(&n, 4, 1, f) != 1 || fread(&budget, 4, 1, f) != 1 || fread(&mi, 4, 1, f) != 1 || n < 1 ||
        n > SM_MAX_N || fread(x, 4, n, f) != (size_t)n || fread(init, 8, 2 * n, f) != (size_t)(2 * n)) {
        fprintf(stderr, "bad sample file\n");
        return 2;
    }
    float tr[SM_MAX_N];
    if (budget < 0 && fread(tr, 4, n - 1, f) != (size_t)(n - 1)) { fprintf(stderr, "bad sample file\n"); return 2; }
    fclose(f);
    const int cap = n + 1, W = SM_WORDS(n), ld = cap + 1;
    uint32_t* ab = calloc((size_t)cap * W, 4);
    float* p = calloc(cap, 4);
    double* p64 = calloc(cap, 8);
    double* inv = calloc((size_t)ld * ld, 8);
    int32_t alen, it, st;
    if (budget < 0) oracle_sparsemap_seq_fwd(x, tr, init, mi, 1, n, ab, p, p64, &alen, inv, &it, &st);
    else oracle_sparsemap_fwd(x, init, budget, mi, 1, n, ab, p, p64, &alen, inv, &it, &st);
    fprintf(stderr, "END alen %d iters %d status %d\n", alen, it, st);
    return 0;
}
