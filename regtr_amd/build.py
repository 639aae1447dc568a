"""Builds regtr_amd/libregtr_hip.so (gfx950) in-tree with hipcc.  `python -m regtr_amd.build [--force] [--experimental]`.
--experimental additionally builds libregtr_hip.experimental.so (-DREGTR_EXPERIMENTAL: the measured-slower experiment kernels of
include/regtr_hip_experimental.h, which the shipped library does not contain; regtr_amd/experimental.py)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libregtr_hip.so')
# the parity-mode neighbour search (include/regtr_hip_parity.h): nanoflann's KD-tree order + std::sort replayed on the GPU -- a checking mode's
# library of its own, so that the product library holds no nanoflann-derived code (THIRD_PARTY_NOTICES.md)
PARITY_LIB = os.path.join(HERE, 'libregtr_parity.so')
PARITY_SOURCES = ['ref_order.hip']
# development only: REGTR_VARIANT=name REGTR_VARIANT_FLAGS='-DX=1' builds libregtr_hip.name.so for A/B kernel experiments
VARIANT = os.environ.get('REGTR_VARIANT', '')
VARIANT_FLAGS = os.environ.get('REGTR_VARIANT_FLAGS', '').split()
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
SOURCES = ['preprocess.hip', 'kpconv.hip', 'gemm.hip', 'gemm_x3.hip', 'gemm_stream.hip', 'block_tail.hip', 'norm.hip', 'attention.hip', 'cross_encoder.hip', 'encoder.hip', 'procrustes.hip']
# bit-level parity of the float32 distance / voxel arithmetic with the reference's SSE2 build needs no contraction
# kpconv.hip: SLP-packed f32 VALU (v_pk_*) beside MFMAs costs more than it saves and blocks v_add_f32_dpp fusion
EXTRA = {'preprocess.hip': ['-ffp-contract=off'], 'ref_order.hip': ['-ffp-contract=off'], 'kpconv.hip': ['-fno-slp-vectorize']}
COMMON = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result',
          '-I' + os.path.join(os.path.dirname(HERE), 'include')]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_hash():
    """sha256 over the kernel sources and headers (sorted by name): the code version of libregtr_hip.so.  profiles/pmc_traffic.json is
    stamped with it and bench.py reports the counter traffic only when it matches the sources the loaded library was built from."""
    import hashlib
    h = hashlib.sha256()
    inc = os.path.join(os.path.dirname(HERE), 'include')
    files = sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h'))]
                   + [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith('.h')])
    for f in files:
        h.update(os.path.basename(f).encode() + b'\0')
        h.update(open(f, 'rb').read())
    return h.hexdigest()


def write_build_info():
    """regtr_amd/_build_info.json next to the library: source hash + git commit (the GPU boxes receive a snapshot without .git)."""
    import json
    info = {'source_sha256': source_hash(), 'git_commit': None, 'git_dirty': None}
    try:
        root = os.path.dirname(HERE)
        info['git_commit'] = subprocess.check_output(['git', '-C', root, 'rev-parse', 'HEAD'], stderr=subprocess.DEVNULL, text=True).strip()
        info['git_dirty'] = bool(subprocess.check_output(['git', '-C', root, 'status', '--porcelain', '--', 'regtr_amd', 'include'],
                                                         stderr=subprocess.DEVNULL, text=True).strip())
    except (OSError, subprocess.CalledProcessError):
        pass
    with open(os.path.join(HERE, '_build_info.json'), 'w') as f:
        json.dump(info, f, indent=1)
    return info


def build(force=False, verbose=False, variant=None, variant_flags=None):
    """variant / variant_flags default to REGTR_VARIANT / REGTR_VARIANT_FLAGS (development builds next to the product library)."""
    VARIANT = globals()['VARIANT'] if variant is None else variant
    VARIANT_FLAGS = globals()['VARIANT_FLAGS'] if variant_flags is None else list(variant_flags)
    LIB = os.path.join(HERE, f'libregtr_hip.{VARIANT}.so') if VARIANT else globals()['LIB']
    objdir = os.path.join(HERE, 'build', VARIANT) if VARIANT else os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    inc = os.path.join(os.path.dirname(HERE), 'include')
    headers = ([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
               + [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith('.h')])      # the public C-ABI header is a dependency too

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + headers):
            cmd = [HIPCC] + COMMON + EXTRA.get(src, []) + VARIANT_FLAGS + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            subprocess.check_call(cmd)
            return o, True
        return o, False

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        res = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    if not VARIANT:
        build_parity(force, verbose)
    if not VARIANT and os.path.isdir(os.path.join(os.path.dirname(HERE), '.git')):
        write_build_info()
    return LIB


def build_parity(force=False, verbose=False):
    """libregtr_parity.so from csrc/ref_order.hip (regtr_amd/_lib.py: parity_lib; loaded only when parity mode asks for KD-tree tables)."""
    objdir = os.path.join(HERE, 'build', 'parity')
    os.makedirs(objdir, exist_ok=True)
    inc = os.path.join(os.path.dirname(HERE), 'include')
    deps = ([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith('.h')])
    objs, changed = [], False
    for src in PARITY_SOURCES:
        s, o = os.path.join(CSRC, src), os.path.join(objdir, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + deps):
            cmd = [HIPCC] + COMMON + EXTRA.get(src, []) + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            subprocess.check_call(cmd)
            changed = True
        objs.append(o)
    if force or changed or not os.path.exists(PARITY_LIB):
        subprocess.check_call([HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', PARITY_LIB] + objs)
    return PARITY_LIB


def build_experimental(force=False, verbose=False):
    """libregtr_hip.experimental.so: the product sources + the experiment kernels (regtr_amd/experimental.py)."""
    return build(force, verbose, variant='experimental', variant_flags=['-DREGTR_EXPERIMENTAL=1'])


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
    print(build_parity(force='--force' in sys.argv, verbose=True))
    if '--experimental' in sys.argv:
        print(build_experimental(force='--force' in sys.argv, verbose=True))
