"""TEST INFRASTRUCTURE ONLY -- imports the REAL reference RegTR from /root/reference (this container
only; the tree does not exist on the GPU box) so that golden vectors can be generated and the
restatement in oracle/regtr_ref.py can be pinned.  Nothing under tests -m gpu, smoke() or bench.py
may call this at run time.

The reference imports a number of packages that are not installed here (MinkowskiEngine, pytorch3d,
open3d, vtk, tensorboard, ...).  None of them is on the CPU inference path once the reference's own
CPU ``Preprocessor`` (kpconv.py:291-414) is used, so they are stubbed in ``sys.modules``.  The
reference's commented-out ``cpp_subsampling`` / ``cpp_neighbors`` imports (kpconv.py:12-15) are
re-created as adapter objects over the unmodified reference C++ (oracle/_ref).
"""
import contextlib
import os
import sys
import types
from unittest import mock

import numpy as np

REF_ROOT = '/root/reference'
REF_SRC = os.path.join(REF_ROOT, 'src')

_STUBS = ['MinkowskiEngine', 'pytorch3d', 'pytorch3d.ops', 'tensorboard', 'torch.utils.tensorboard',
          'nibabel', 'nibabel.quaternions', 'open3d', 'vtk', 'vtk.util', 'vtk.util.numpy_support',
          'coloredlogs', 'git', 'h5py', 'torchvision', 'torchvision.transforms', 'matplotlib',
          'matplotlib.pyplot', 'matplotlib.cm', 'matplotlib.colors', 'pandas']


def available():
    return os.path.isdir(REF_SRC)


class _EasyDict(dict):
    """Minimal easydict.EasyDict stand-in (attribute access + dict protocol)."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


class _CppSubsampling:
    """cpp_subsampling.subsample_batch(points, batches, sampleDl=, max_p=, verbose=)
    -- cpp_subsampling/wrapper.cpp:75,322."""
    @staticmethod
    def subsample_batch(points, batches, features=None, classes=None, sampleDl=0.1, method='barycenters',
                        max_p=0, verbose=0):
        from oracle import native
        assert features is None and classes is None
        return native.ref_subsample_batch(np.asarray(points), np.asarray(batches), sampleDl, max_p)


class _CppNeighbors:
    """cpp_neighbors.batch_query(queries, supports, q_batches, s_batches, radius=)
    -- cpp_neighbors/wrapper.cpp:71-75,214-227."""
    @staticmethod
    def batch_query(queries, supports, q_batches, s_batches, radius=0.1):
        from oracle import native
        return native.ref_batch_query(np.asarray(queries), np.asarray(supports), np.asarray(q_batches),
                                      np.asarray(s_batches), radius)


_loaded = None


def load():
    """Import the reference package tree; returns a namespace with the modules of interest."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError('reference tree not present')
    for name in _STUBS:
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock(name=name)
    ed = types.ModuleType('easydict')
    ed.EasyDict = _EasyDict
    sys.modules.setdefault('easydict', ed)
    sys.path.insert(0, REF_SRC)
    with chdir(REF_SRC):
        import models.regtr as regtr                       # noqa
        import models.backbone_kpconv.kpconv as kpconv     # noqa
        import models.backbone_kpconv.kpconv_blocks as kpconv_blocks  # noqa
        import models.transformer.transformers as transformers        # noqa
        import models.transformer.position_embedding as position_embedding  # noqa
        import utils.se3_torch as se3_torch                # noqa
        import utils.misc as misc                          # noqa
    kpconv.cpp_subsampling = _CppSubsampling
    kpconv.cpp_neighbors = _CppNeighbors
    _loaded = types.SimpleNamespace(regtr=regtr, kpconv=kpconv, kpconv_blocks=kpconv_blocks,
                                    transformers=transformers, position_embedding=position_embedding,
                                    se3_torch=se3_torch, misc=misc, EasyDict=_EasyDict)
    return _loaded


@contextlib.contextmanager
def chdir(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


def load_cfg(name):
    """name in {'3dmatch','modelnet'} -> EasyDict of the flattened reference YAML (utils/misc.py:10-29)."""
    ref = load()
    return _EasyDict(ref.misc.load_config(os.path.join(REF_SRC, 'conf', f'{name}.yaml')))


def build_model(cfg, seed=0):
    """Seeded random-init reference RegTR on CPU with the reference's CPU Preprocessor
    (kpconv.py:291) swapped in for PreprocessorGPU (regtr.py:29)."""
    import torch
    ref = load()
    torch.manual_seed(seed)
    np.random.seed(seed)          # kernel-point rotation/noise: kernel_points.py:434-461
    with chdir(REF_SRC):          # load_kernels opens the cwd-relative 'kernels/dispositions' (:390)
        model = ref.regtr.RegTR(cfg)
    model.preprocessor = ref.kpconv.Preprocessor(cfg)
    model.eval()
    return model
