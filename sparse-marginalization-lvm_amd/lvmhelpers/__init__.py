"""MI355X-native drop-in for the hot path of ``lvmhelpers`` (deep-spin/sparse-marginalization-lvm).

Same module paths, class names, constructor/forward signatures and ``log`` keys as the reference
package for the sparse-marginalization path:

    lvmhelpers.marg        entropy, ExplicitWrapper, Marginalizer
    lvmhelpers.structmarg  TopKSparsemaxWrapper, TopKSparsemaxMarg, SparseMAPWrapper, SparseMAPMarg
    lvmhelpers.sparsemap   bernoulli_smap, budget_smap
    lvmhelpers.pbinary_topk / pbernoulli / pbudget   the former Cython oracles
    lvmhelpers.utils       DeterministicWrapper, populate_common_params

The arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI of
``include/sparsemarg.h`` (``lvmhelpers._native``); there is no CPU fallback.
"""
