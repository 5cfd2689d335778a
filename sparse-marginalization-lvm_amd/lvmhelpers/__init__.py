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

Fall-through package: the modules above are the hot path and live here; everything else of the
reference's ``lvmhelpers`` -- the sampling baselines ``sfe``, ``nvil``, ``gumbel``,
``sum_and_sample``, which the experiment scripts import at module top
(experiments/signal-game/train.py:11-18, bit_vector-vae/train.py:13-19,
semi_supervised-vae/train.py:13-20) -- is not rebuilt and resolves from a reference checkout that
comes LATER on ``sys.path``: ``PYTHONPATH=<this package dir>:<reference checkout>``.  A name that
exists in both places is taken from here, because this directory is first on ``__path__``.
"""
import pkgutil

__path__ = pkgutil.extend_path(__path__, __name__)
