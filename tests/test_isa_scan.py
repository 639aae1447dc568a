"""CPU: tools/isa_scan.py -- the scanner that found the loads hipcc had serialised (load immediately followed by s_waitcnt vmcnt(0))
in the round-3 kernels.  Unit test on assembly text, plus a smoke run over one real source file."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

ASM = '''
	.text
_Z10serialisedPKfPf:                    ; @_Z10serialisedPKfPf
; %bb.0:
	s_load_dwordx4 s[0:3], s[4:5], 0x0
	v_mov_b32_e32 v1, 0
	s_and_saveexec_b64 s[6:7], vcc
	s_cbranch_execz .LBB0_2
	global_load_dwordx4 v[2:5], v0, s[0:1]
	s_waitcnt vmcnt(0)
.LBB0_2:
	s_or_b64 exec, exec, s[6:7]
	s_and_saveexec_b64 s[6:7], vcc
	global_load_dwordx4 v[6:9], v0, s[0:1] offset:16
	v_add_u32_e32 v1, 1, v1
	s_waitcnt vmcnt(0)
	global_load_lds_dwordx4 v0, s[0:1]
	s_waitcnt vmcnt(0)
	s_endpgm
_Z8batchedPKfPf:                        ; @_Z8batchedPKfPf
	global_load_dwordx4 v[2:5], v0, s[0:1]
	global_load_dwordx4 v[6:9], v0, s[0:1] offset:16
	buffer_load_dword v10, v0, s[0:3], 0 offen
	v_add_u32_e32 v1, 1, v1
	v_add_u32_e32 v1, 1, v1
	v_add_u32_e32 v1, 1, v1
	s_waitcnt vmcnt(0)
	s_endpgm
'''


def test_scan_counts_loads_followed_by_a_full_wait():
    import isa_scan
    rows = {name: (loads, stalled) for name, loads, stalled, _ in isa_scan.scan(ASM, 3)}
    assert rows['_Z10serialisedPKfPf'] == (2, 2)        # the LDS-DMA load is not counted: it is waited for by design
    assert rows['_Z8batchedPKfPf'] == (3, 0)             # last load: the wait is outside the 3-instruction window


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='hipcc not installed')
def test_scan_runs_on_a_kernel_source(capsys):
    import isa_scan
    sys.argv = ['isa_scan.py', os.path.join(ROOT, 'regtr_amd', 'csrc', 'norm.hip'), '--min-stalled', '0']
    isa_scan.main()
    out = capsys.readouterr().out
    assert 'k_instnorm_apply' in out and 'k_layernorm' in out
