"""MI355X-native RegTR correspondence-inference hot path (gfx950 HIP kernels behind a C-ABI).

Public surface mirrors the reference (citations relative to /root/reference/src):
  RegTR(cfg).forward(batch)           models/regtr.py:22,104
  cpp_wrappers.subsample_batch / batch_query   models/backbone_kpconv/cpp_wrappers/*
  compute_rigid_transform             utils/se3_torch.py:108
  overlap.compute_overlap / compute_overlaps   utils/pointcloud.py:8, models/backbone_kpconv/kpconv.py:540
cfg keys beyond the reference's: kpconv_ref_row_order (parity mode), compute_dtype ('fp32' | 'bf16' | 'bf16x2'), kpconv_meta_int64.
"""
__all__ = ['RegTR', 'load_config']


def __getattr__(name):
    if name == 'RegTR':
        from .regtr import RegTR
        return RegTR
    if name == 'load_config':
        from .config import load_config
        return load_config
    raise AttributeError(name)
