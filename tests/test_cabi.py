"""CPU: the C-ABI library loads and exports exactly what include/regtr_hip.h declares; host-side contracts."""
import os
import re

import numpy as np
import pytest
import torch

from tests.util import ROOT, load_cfg, seeded_sd


def _header_symbols(name='regtr_hip.h'):
    txt = open(os.path.join(ROOT, 'include', name)).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(regtr_\w+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from regtr_amd import _lib
    lib = _lib.lib()
    names = _header_symbols()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/regtr_hip.h but not exported'
    assert sorted(_lib.SIGNATURES) == names, 'ctypes signatures and header disagree'
    # the shipped library exports EXACTLY the drop-in boundary: the measured-slower experiment kernels (include/regtr_hip_experimental.h)
    # are compiled only into libregtr_hip.experimental.so (regtr_amd/experimental.py), outside the ABI version
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r'\bT (regtr_\w+)', out)))
    assert exported == names, set(exported) ^ set(names)
    # ... and NOTHING else that is code or a kernel handle: every kernel lives in an anonymous namespace / is static (round 5 shipped
    # `__device_stub__k_add` and the k_maxpool_gather_buf<> stubs beside the ABI).  Left over: the HIP fat-binary bookkeeping symbols.
    stray = [l.split()[-1] for l in out.splitlines() if l.split() and not re.match(r'regtr_\w+$|__hip_cuid_\w+$|_(_)?(init|fini|edata|end|bss_start)$', l.split()[-1])]
    assert not stray, f'libregtr_hip.so exports symbols outside include/regtr_hip.h: {stray}'
    from regtr_amd import experimental
    exp = _header_symbols('regtr_hip_experimental.h')
    assert sorted(experimental.SIGNATURES) == exp and not set(exp) & set(names)
    if experimental.available():
        out = subprocess.run(['nm', '-D', '--defined-only', experimental.LIB_PATH], capture_output=True, text=True, check=True).stdout
        assert sorted(set(re.findall(r'\bT (regtr_\w+)', out))) == sorted(names + exp)
    txt = open(os.path.join(ROOT, 'include', 'regtr_hip.h')).read()
    assert int(re.search(r'#define REGTR_ABI_VERSION (\d+)', txt).group(1)) == _lib.ABI_VERSION == lib.regtr_abi_version()
    # the parity-mode library (include/regtr_hip_parity.h): its own header, its own exports, the same ABI version -- and no KD-tree symbol
    # in the product library (the nanoflann-derived code of csrc/ref_kdtree.h ships in libregtr_parity.so only)
    par = _header_symbols('regtr_hip_parity.h')
    assert sorted(_lib.PARITY_SIGNATURES) == par and not set(par) & set(names) and not [n for n in names if 'kdtree' in n]
    pl = _lib.parity_lib()
    assert pl.regtr_parity_abi_version() == _lib.ABI_VERSION
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.PARITY_LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert sorted(set(re.findall(r'\bT (regtr_\w+)', out))) == par


def test_stray_env_switch_does_not_reroute():
    """The A/B switches of the dispatch are read from the environment only under REGTR_DEV=1 (regtr_amd/devflags.py): a stray REGTR_*
    variable in a production environment changes nothing; and the experiment kernels have no route in the product at all."""
    import subprocess
    import sys
    code = ('import regtr_amd.ops as o, regtr_amd.regtr as r, regtr_amd._lib as l;'
            'print(o.f16_pair_default, o.use_stream_gemm, o.use_block_tail, o.prenorm_gather, o.use_tile_info, o.preapply_unary2,'
            ' o.use_one_call_cross_encoder, r.overlap_preprocessing, l.LIB_PATH.endswith("libregtr_hip.so"),'
            ' hasattr(o, "use_fused_kpconv"), hasattr(o, "block_tail_res"), hasattr(o, "thin_f16_gemm"))')
    stray = {'REGTR_FUSED_KPCONV': '1', 'REGTR_BLOCK_TAIL_RES': '1', 'REGTR_F16_THIN': '1', 'REGTR_F16_PAIR': '0', 'REGTR_STREAM_GEMM': '0',
             'REGTR_BLOCK_TAIL': '0', 'REGTR_PRENORM': '0', 'REGTR_TILE_INFO': '0', 'REGTR_PREAPPLY_UNARY2': '0', 'REGTR_ONE_CALL_XENC': '0',
             'REGTR_OVERLAP': '0', 'REGTR_VARIANT': 'nosuch'}
    env = {k: v for k, v in os.environ.items() if not k.startswith('REGTR_')}
    run = lambda e: subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=e, capture_output=True, text=True, check=True).stdout.split()
    clean = run(env)
    assert clean == ['True', 'True', 'True', 'True', 'True', '1', 'True', 'True', 'True', 'False', 'False', 'False']
    assert run(dict(env, **stray)) == clean                                   # stray variables: no routing change
    dev = run(dict(env, REGTR_DEV='1', **{k: v for k, v in stray.items() if k != 'REGTR_VARIANT'}))
    assert dev[:8] == ['False', 'False', 'False', 'False', 'False', '0', 'False', 'False'] and dev[9:] == ['False'] * 3


def test_context_is_thread_local():
    """The per-forward state (launch device, operand format, status word, timing lists) lives in a thread-local context stack
    (regtr_amd/context.py), not in module globals: what one host thread sets, another does not see."""
    import threading
    from regtr_amd import context, ops
    seen = {}
    gate_a, gate_b = threading.Event(), threading.Event()

    def worker():
        seen['before'] = context.current().f16_pair
        with ops.f16_pair(False):
            gate_a.set(); gate_b.wait(5)
            seen['inside'] = context.current().f16_pair
    t = threading.Thread(target=worker)
    with ops.f16_pair(True), context.recording(gather_records=[1]):
        t.start(); gate_a.wait(5)
        assert context.current().f16_pair is True and context.current().gather_records == [1]
        with ops.f16_pair(False):
            assert context.current().f16_pair is False and context.current().gather_records == [1]      # nested: fields inherited
        assert context.current().f16_pair is True
        gate_b.set(); t.join()
    assert seen == {'before': False, 'inside': False}
    assert context.current().f16_pair is False and context.current().gather_records is None


def test_workspace_size_queries_are_host_only():
    from regtr_amd import _lib
    lib = _lib.lib()
    a, b = lib.regtr_grid_subsample_ws_bytes(1000, 2), lib.regtr_grid_subsample_ws_bytes(100000, 2)
    assert 0 < a < b
    assert lib.regtr_cellgrid_ws_bytes(50000, 4) > lib.regtr_cellgrid_ws_bytes(500, 4) > 0
    assert lib.regtr_instnorm_ws_bytes(2, 20000, 128) >= 2 * 79 * 128 * 16


def test_argument_errors_return_status_not_crash():
    from regtr_amd import _lib
    lib = _lib.lib()
    assert lib.regtr_gemm_f32(None, 4, None, 4, None, 4, 1, 4, 4, None, None, None, 0, 0, None, None, 0, 0.1, None, 0, None) == -2
    assert lib.regtr_gemm_f32_ws_bytes(100000, 128, 64) == 0 and lib.regtr_gemm_f32_ws_bytes(751, 256, 3840) > 0
    assert lib.regtr_radius_query(None, None, 1, None, 1, 1, 0.1, 40, 0, None, 0, None, None, None, None) == -2
    assert lib.regtr_grid_subsample(None, None, 1, 1, 0.1, None, None, None, 0, None) == -2
    with pytest.raises(RuntimeError):
        _lib.check(-3, 'x')


@pytest.mark.parametrize('name', ['3dmatch', 'modelnet'])
def test_state_dict_contract(name):
    """Parameter names / shapes equal the reference's (cross-checked against the real reference
    module in oracle/make_golden.py) and a reference-style checkpoint loads strictly (demo.py:165)."""
    from regtr_amd import RegTR
    from oracle.seeded_weights import param_shapes
    cfg = load_cfg(name)
    m = RegTR(cfg)
    sd = m.state_dict()
    ref = param_shapes(cfg)
    assert list(sd.keys()) == list(ref.keys())
    for k, s in ref.items():
        assert tuple(sd[k].shape) == tuple(s), k
    m.load_state_dict(seeded_sd(cfg), strict=True)
    assert sum(p.numel() for p in m.parameters()) == (11845811 if name == '3dmatch' else 11488018)


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU instead of silently computing somewhere else."""
    from regtr_amd import RegTR, ops
    cfg = load_cfg('modelnet')
    m = RegTR(cfg)
    batch = {'src_xyz': [torch.rand(50, 3)], 'tgt_xyz': [torch.rand(60, 3)]}
    with pytest.raises(RuntimeError):
        m(batch)
    with pytest.raises(RuntimeError):
        ops.gemm(torch.rand(4, 4), torch.rand(4, 4))


def test_product_never_imports_oracle():
    import ast
    pkg = os.path.join(ROOT, 'regtr_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            tree = ast.parse(open(os.path.join(pkg, fn)).read())
            for node in ast.walk(tree):
                mods = []
                if isinstance(node, ast.Import):
                    mods = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    mods = [node.module or '']
                assert not any(m.split('.')[0] == 'oracle' for m in mods), f'{fn} imports oracle'


def test_config_contract(tmp_path):
    from regtr_amd.config import load_config
    p = tmp_path / 'c.yaml'
    p.write_text('a:\n  x: 1\n  y: [1, 2]\nb:\n  z: hello\n')
    cfg = load_config(str(p))
    assert cfg.x == 1 and cfg['y'] == [1, 2] and cfg.get('q', 5) == 5 and 'z' in cfg   # utils/misc.py:24-27 flattening


def test_kernel_points_loader_matches_reference_recipe():
    from regtr_amd.kernel_points import K015_CENTER, load_kernels
    np.random.seed(0)
    kp = load_kernels(0.0625, 15, 3, 'center')
    assert kp.shape == (15, 3) and kp.dtype == np.float32
    np.random.seed(0)
    th = np.random.rand() * 2 * np.pi
    c, s = np.cos(th), np.sin(th)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)
    exp = np.matmul(0.0625 * (K015_CENTER + np.random.normal(scale=0.01, size=(15, 3))), R).astype(np.float32)
    assert np.array_equal(kp, exp)
    assert np.allclose(np.linalg.norm(K015_CENTER[1:], axis=1).mean(), 0.66, atol=0.02)


def test_reference_order_bare_call_warns_and_setter_is_thread_local():
    """`cpp_wrappers.reference_order(True)` used to be a setter; as a context manager a bare call changes nothing, so it must say
    so; `set_reference_order` is the setter, and it touches the calling thread only."""
    import gc
    import threading
    import warnings
    from regtr_amd import context, cpp_wrappers
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        cpp_wrappers.reference_order(True)          # never entered
        gc.collect()
    assert any('never entered' in str(x.message) for x in w)
    assert not context.current().reference_order
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        with cpp_wrappers.reference_order():
            assert context.current().reference_order
        gc.collect()
    assert not w and not context.current().reference_order
    seen = []
    def other():
        seen.append(context.current().reference_order)
    assert cpp_wrappers.set_reference_order(True) is False and context.current().reference_order
    t = threading.Thread(target=other); t.start(); t.join()
    assert seen == [False]
    assert cpp_wrappers.set_reference_order(False) is True and not context.current().reference_order
