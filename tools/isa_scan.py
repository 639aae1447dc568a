"""Development tool: two compiler pitfalls found in this library's kernels, read from the ISA (no GPU needed).

(1) kernels whose global loads are serialised by the compiler:

hipcc turns a predicated load (`x = 0; if (ok) x = p[i];`, `ok ? p[i] : 0`, a load under `if (more)` into a loop-carried
register) into  branch / load / s_waitcnt vmcnt(0)  -- the wave sleeps a full memory round trip PER LOAD instead of keeping
them all in flight (measured on MI355X: attention K/V fetch 173 -> 148 us per launch, max-pool gather 887 -> 626 us once the
loads were made branch-free from clamped rows / range-checked buffer loads).  This prints, per kernel of the given .hip files,
how many of its vector-memory loads are followed (within `--window` instructions, before any other load) by a vmcnt(0) wait.

(2) kernels whose MFMA accumulators live in AGPRs (`--agpr`): without an occupancy hint a 256-thread kernel has a 512-register budget
and hipcc parks MFMA results in AGPRs; every vector-ALU use of them costs a v_accvgpr_read / _write (attention: 288 of 1151 vector
instructions, 160 registers instead of 124).  `__attribute__((amdgpu_waves_per_eu(N)))` with a budget of at most 256 registers keeps them in VGPRs.

    python tools/isa_scan.py regtr_amd/csrc/gemm.hip regtr_amd/csrc/norm.hip        (hipcc -S for gfx950)
    python tools/isa_scan.py --agpr regtr_amd/csrc/*.hip
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def scan(asm, window):
    rows = []
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', asm, re.M | re.S):
        name, body = m.group(1), m.group(2)
        ins = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith((';', '.', '//'))]
        ins = [l for l in ins if not l.endswith(':')]
        loads = stalled = 0
        for i, l in enumerate(ins):
            if not l.startswith(('global_load', 'buffer_load', 'flat_load')) or ' lds' in l or '_lds_' in l.split()[0]:
                continue        # (LDS-DMA loads are waited for by count, by design)
            loads += 1
            for nxt in ins[i + 1:i + 1 + window]:
                if nxt.startswith(('global_load', 'buffer_load', 'flat_load')):
                    break
                if nxt.startswith('s_waitcnt') and 'vmcnt(0)' in nxt:
                    stalled += 1
                    break
        rows.append((name, loads, stalled, len(ins)))
    return rows


def agpr_copies(asm):
    """(kernel, v_accvgpr_* instructions, MFMAs, vector instructions) per kernel that has any MFMA."""
    rows = []
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', asm, re.M | re.S):
        ins = re.findall(r'^\s+([a-z_0-9]+)', m.group(2), re.M)
        mf = sum(i.startswith('v_mfma') for i in ins)
        if mf:
            rows.append((m.group(1), sum(i.startswith('v_accvgpr') for i in ins), mf, sum(i.startswith('v_') for i in ins)))
    return rows


def structure(asm, needle):
    """One-line map of a kernel: L load, D lds-dma load, W<n> vmcnt wait, S store, M mfma, BAR barrier, | branch target."""
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', asm, re.M | re.S):
        if needle not in m.group(1):
            continue
        ev = []
        for l in (x.strip() for x in m.group(2).split('\n')):
            if not l or l.startswith((';', '//')):
                continue
            if l.endswith(':') and l.startswith('.LBB'):
                ev.append('|')
            elif l.startswith(('global_load', 'buffer_load', 'flat_load')):
                ev.append('D' if (' lds' in l or '_lds_' in l.split()[0]) else 'L')
            elif l.startswith('s_waitcnt') and 'vmcnt(' in l:
                ev.append('W' + l.split('vmcnt(')[1].split(')')[0])
            elif l.startswith(('global_store', 'buffer_store')):
                ev.append('S')
            elif l.startswith('v_mfma'):
                ev.append('M')
            elif l.startswith('s_barrier'):
                ev.append('BAR')
            elif l.startswith('s_cbranch'):
                ev.append('br')
        out, prev, cnt = [], None, 0
        for e in ev + [None]:
            if e == prev:
                cnt += 1
                continue
            if prev is not None:
                out.append(prev if cnt == 1 else f'{prev}x{cnt}')
            prev, cnt = e, 1
        print(m.group(1)[:100])
        print('   ' + ' '.join(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('sources', nargs='+')
    ap.add_argument('--window', type=int, default=3)
    ap.add_argument('--min-stalled', type=int, default=2)
    ap.add_argument('--agpr', action='store_true', help='report v_accvgpr_* copies per MFMA kernel instead')
    ap.add_argument('--map', default=None, help='print the load/wait/branch structure of kernels whose mangled name contains this')
    args = ap.parse_args()
    for src in args.sources:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, 'k.s')
            extra = ['-fno-slp-vectorize'] if os.path.basename(src) == 'kpconv.hip' else []      # as regtr_amd/build.py compiles it
            cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-I', os.path.join(ROOT, 'include'),
                   '-I', os.path.join(ROOT, 'regtr_amd', 'csrc'), src, '-o', out] + extra
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                sys.exit(r.stderr[-2000:])
            asm = open(out).read()
            if args.map:
                structure(asm, args.map)
                continue
            if args.agpr:
                rows = agpr_copies(asm)
                names = demangle([r[0] for r in rows])
                print(f'== {src}: {sum(1 for r in rows if r[1])} of {len(rows)} MFMA kernels copy through AGPRs')
                for name, acc, mf, valu in sorted(rows, key=lambda r: -r[1]):
                    if acc:
                        print(f'  {acc:4d} v_accvgpr_* of {valu:5d} vector instructions, {mf:3d} MFMAs  {names[name][:110]}')
                continue
            rows = scan(asm, args.window)
        if args.map or args.agpr:
            continue
        names = demangle([r[0] for r in rows])
        print(f'== {src}')
        for name, loads, stalled, n in sorted(rows, key=lambda r: -r[2]):
            if stalled >= args.min_stalled:
                print(f'  {stalled:4d} of {loads:4d} loads wait at once  ({n:5d} instr)  {names[name][:110]}')


if __name__ == '__main__':
    main()
