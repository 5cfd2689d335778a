// Diagnostic: cycles per phase of the SparseMAP solver (never shipped; build:
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSMLVM_SM_PROF tools/sm_prof.hip -o tools/sm_prof;
// without -DSMLVM_SM_PROF (-o tools/sm_time) it only times the launch, for rocprofv3 --pmc runs)
#include "../sparse-marginalization-lvm_amd/csrc/sparsemap.hip"
#include <cstdio>
#include <vector>
#include <random>
namespace smlvm { thread_local int g_last_hip_error = 0; }
int main(int argc, char** argv)
{
    int B = argc > 1 ? atoi(argv[1]) : 1024, n = argc > 2 ? atoi(argv[2]) : 128, budget = argc > 3 ? atoi(argv[3]) : 0;
    float sigma = argc > 4 ? atof(argv[4]) : 1.0f;
    std::mt19937 g(1);
    std::normal_distribution<float> nd(0, sigma);
    std::uniform_real_distribution<double> ud(0, 1);
    std::vector<float> x((size_t)B * n);
    std::vector<double> init((size_t)B * 2 * n);
    for (auto& v : x) v = nd(g);
    for (auto& v : init) v = ud(g);
    int cap = n + 1, W = (n + 31) / 32, ld = n + 2;
    float *dx, *dp; double *dinit, *dinv; uint32_t* dbits; int *dalen, *dit, *dst;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dinit, init.size() * 8); hipMalloc(&dp, (size_t)B * cap * 4);
    hipMalloc(&dbits, (size_t)B * cap * W * 4); hipMalloc(&dalen, B * 4); hipMalloc(&dit, B * 4); hipMalloc(&dst, B * 4);
    hipMalloc(&dinv, (size_t)B * ld * ld * 8);
    size_t wsb = smlvm_sparsemap_fwd_workspace_bytes(B, n, 1); void* dws = nullptr; hipMalloc(&dws, wsb + 8);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dinit, init.data(), init.size() * 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
#ifdef SMLVM_SM_PROF
        static unsigned long long* dprof = nullptr;
        if (!dprof) {
            hipMalloc(&dprof, (size_t)B * 16 * 8);
            hipMemcpyToSymbol(HIP_SYMBOL(smlvm::g_prof_out), &dprof, sizeof dprof);
        }
        hipMemset(dprof, 0, (size_t)B * 16 * 8);
#endif
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        int rc = smlvm_sparsemap_fwd(dx, dinit, budget, 300, B, n, dbits, dp, dalen, dinv, dit, dst, dws, wsb, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<int> it(B); hipMemcpy(it.data(), dit, B * 4, hipMemcpyDeviceToHost);
        long long tot = 0; int itmax = 0; for (int v : it) { tot += v; itmax = v > itmax ? v : itmax; }
        printf("rc=%d  %.3f ms  total iters %lld  max %d\n", rc, ms, tot, itmax);
#ifdef SMLVM_SM_PROF
        std::vector<unsigned long long> pr((size_t)B * 16);
        hipMemcpy(pr.data(), dprof, pr.size() * 8, hipMemcpyDeviceToHost);
        const char* names[] = {"solve z", "marginals", "maximize+eval", "dup+r+push", "d=invA r", "sden+t1", "A update+append", "C7 polish", "blocking(no)", "blocking+removal"};
        for (int pass = 0; pass < 2; ++pass) {   // all samples, then only the ones that ran longest
            unsigned long long prof[16] = {0}; long long its = 0; int cnt = 0;
            for (int b = 0; b < B; ++b) {
                if (pass == 1 && it[b] < itmax) continue;
                for (int k = 0; k < 16; ++k) prof[k] += pr[(size_t)b * 16 + k];
                its += it[b]; ++cnt;
            }
            unsigned long long sum = 0; for (int k = 0; k < 10; ++k) sum += prof[k];
            printf(" %s: %d samples, %lld passes, %.0f ticks per pass, %.0f ticks per sample\n", pass ? "longest" : "all", cnt, its, (double)sum / its, (double)sum / cnt);
            for (int k = 0; k < 10; ++k) printf("  %-18s %14llu ticks  %5.1f%%\n", names[k], prof[k], 100.0 * prof[k] / sum);
        }
#endif
    }
    return 0;
}
