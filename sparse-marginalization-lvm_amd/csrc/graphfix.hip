// Repair pass over a captured HIP graph: every memset node becomes a kernel node.
//
// No counterpart in the reference.  Why it exists: with the HIP runtime that ships inside this
// image's PyTorch wheel (7.0.51831), a memset captured into a graph writes the captured value on
// the first launch of the instantiated graph only -- from the second replay on the node fills its
// destination with an arbitrary word (graph [memset buf = 0][out = buf]: 0, then -31456768, ...).  PyTorch's own reductions are hit (a column sum of a
// 640 x 3920 matrix zeroes its semaphores with cudaMemsetAsync: 19 of 20 replays wrong,
// tools/graph_memset_probe.py), and so was this library's single-pass compaction (ticket counter)
// before it zeroed its workspace with a kernel.  lvmhelpers/graphs.py keeps the captured
// hipGraph_t, runs this pass over it and only then instantiates it.
#include "common.hpp"

#include <cstring>
#include <vector>

namespace smlvm {

// `width` elements of `elem` (1, 2, 4) bytes per row, `height` rows `pitch` bytes apart
__global__ __launch_bounds__(256) void graph_fill_kernel(unsigned char* __restrict__ dst, unsigned value,
                                                         int elem, unsigned long long width,
                                                         unsigned long long height, unsigned long long pitch)
{
    const unsigned long long total = width * height;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (unsigned long long)gridDim.x * 256) {
        const unsigned long long r = i / width, c = i - r * width;
        unsigned char* row = dst + r * pitch;
        if (elem == 4) reinterpret_cast<unsigned*>(row)[c] = value;
        else if (elem == 2) reinterpret_cast<unsigned short*>(row)[c] = (unsigned short)value;
        else row[c] = (unsigned char)value;
    }
}

static int fix_graph(hipGraph_t g, int* seen, int* replaced, int depth)
{
    if (depth > 8) return SMLVM_ERR_UNSUPPORTED;
    size_t n = 0;
    if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) return SMLVM_ERR_HIP;
    std::vector<hipGraphNode_t> nodes(n);
    if (n && hipGraphGetNodes(g, nodes.data(), &n) != hipSuccess) return SMLVM_ERR_HIP;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType type;
        if (hipGraphNodeGetType(nodes[i], &type) != hipSuccess) return SMLVM_ERR_HIP;
        if (type == hipGraphNodeTypeGraph) {
            hipGraph_t child;
            if (hipGraphChildGraphNodeGetGraph(nodes[i], &child) != hipSuccess) return SMLVM_ERR_HIP;
            const int rc = fix_graph(child, seen, replaced, depth + 1);
            if (rc != SMLVM_OK) return rc;
            continue;
        }
        if (type != hipGraphNodeTypeMemset) continue;
        ++*seen;
        hipMemsetParams mp;
        if (hipGraphMemsetNodeGetParams(nodes[i], &mp) != hipSuccess) return SMLVM_ERR_HIP;
        if (!(mp.elementSize == 1 || mp.elementSize == 2 || mp.elementSize == 4)) return SMLVM_ERR_UNSUPPORTED;
        size_t nd = 0, nt = 0;
        if (hipGraphNodeGetDependencies(nodes[i], nullptr, &nd) != hipSuccess) return SMLVM_ERR_HIP;
        std::vector<hipGraphNode_t> deps(nd);
        if (nd && hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd) != hipSuccess) return SMLVM_ERR_HIP;
        if (hipGraphNodeGetDependentNodes(nodes[i], nullptr, &nt) != hipSuccess) return SMLVM_ERR_HIP;
        std::vector<hipGraphNode_t> outs(nt);
        if (nt && hipGraphNodeGetDependentNodes(nodes[i], outs.data(), &nt) != hipSuccess) return SMLVM_ERR_HIP;

        unsigned char* dst = static_cast<unsigned char*>(mp.dst);
        unsigned value = mp.value;
        int elem = (int)mp.elementSize;
        unsigned long long width = mp.width, height = mp.height ? mp.height : 1, pitch = mp.pitch;
        if (height == 1) pitch = 0;
        if (width == 0) { width = 0; height = 0; }
        const unsigned long long total = width * height;
        unsigned long long blocks = (total + 255) / 256;
        if (blocks < 1) blocks = 1;
        if (blocks > 4096) blocks = 4096;
        void* args[] = {&dst, &value, &elem, &width, &height, &pitch};
        hipKernelNodeParams kp;
        std::memset(&kp, 0, sizeof(kp));
        kp.func = reinterpret_cast<void*>(graph_fill_kernel);
        kp.gridDim = dim3((unsigned)blocks);
        kp.blockDim = dim3(256);
        kp.sharedMemBytes = 0;
        kp.kernelParams = args;
        kp.extra = nullptr;
        hipGraphNode_t knode;
        if (hipGraphAddKernelNode(&knode, g, nd ? deps.data() : nullptr, nd, &kp) != hipSuccess) return SMLVM_ERR_HIP;
        for (size_t t = 0; t < nt; ++t)
            if (hipGraphAddDependencies(g, &knode, &outs[t], 1) != hipSuccess) return SMLVM_ERR_HIP;
        if (hipGraphDestroyNode(nodes[i]) != hipSuccess) return SMLVM_ERR_HIP;
        ++*replaced;
    }
    return SMLVM_OK;
}

}  // namespace smlvm

extern "C" int smlvm_graph_memsets_to_kernels(void* graph, int32_t* n_memsets, int32_t* n_replaced)
{
    if (!graph) return SMLVM_ERR_INVALID;
    int seen = 0, replaced = 0;
    const int rc = smlvm::fix_graph(static_cast<hipGraph_t>(graph), &seen, &replaced, 0);
    if (n_memsets) *n_memsets = seen;
    if (n_replaced) *n_replaced = replaced;
    return rc;
}
