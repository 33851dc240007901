"""MI355X-native Chebyshev graph convolution for fMRI decoding.

Drop-in for the hot path of zhangyu2ustc/GCN_fmri_decoding: ``lib_new.models_gcn.cgcnn``
(``chebyshev5`` / ``b1relu`` / ``b2relu`` / ``mpool1``), ``lib_new.graph`` (Laplacians)
and ``lib_new.coarsening`` (pooling index maps), on PyTorch-ROCm tensors, with the
arithmetic in hand-written gfx950 kernels behind the C ABI of ``include/chebgcn.h``.
"""
from . import _lib  # noqa: F401
from . import graph, coarsening  # noqa: F401

__all__ = ['graph', 'coarsening', 'models_gcn', 'ops']


def __getattr__(name):
    # torch-dependent modules are imported lazily so that the host-only parts
    # (graph, coarsening) stay importable in minimal environments
    if name in ('ops', 'models_gcn', 'dist'):
        import importlib
        return importlib.import_module('.' + name, __name__)
    raise AttributeError(name)
