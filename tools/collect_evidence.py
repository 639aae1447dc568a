"""Copy the summaries of one tools/evidence.sh run (gpurun_out/<tag>/) into profiles/<tag>_* and, with --replace OLD, remove the
profiles/OLD_* set the same script produced earlier (the git history keeps it):
    python tools/collect_evidence.py r04_y --replace r04_z"""
import argparse
import glob
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EVIDENCE_NAMES = ('bench', 'forward_trace', 'host_profile_pairs1', 'kernel_stats', 'pmc_mfma', 'pmc_traffic_kernels', 'pytest_tail', 'e2e_harness')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('tag')
    ap.add_argument('--replace', default='')
    a = ap.parse_args()
    src, dst = os.path.join(ROOT, 'gpurun_out', a.tag), os.path.join(ROOT, 'profiles')
    if a.replace:
        for f in glob.glob(os.path.join(dst, a.replace + '_*')):
            if os.path.basename(f)[len(a.replace) + 1:].startswith(EVIDENCE_NAMES):
                os.remove(f)
    for f in sorted(glob.glob(os.path.join(src, 'bench*.json'))):
        shutil.copy(f, os.path.join(dst, f'{a.tag}_{os.path.basename(f)}'))
    for name, out in (('forward_trace.md', 'forward_trace.md'), ('kernel_stats.md', 'kernel_stats.md'), ('pmc_mfma.md', 'pmc_mfma.md'),
                      ('forward_trace_pairs1.md', 'forward_trace_pairs1.md'), ('kernel_stats_pairs1.md', 'kernel_stats_pairs1.md'),
                      ('forward_trace_real.md', 'forward_trace_real.md'), ('kernel_stats_real.md', 'kernel_stats_real.md'),
                      ('kernel_stats_concurrent.md', 'kernel_stats_concurrent.md'), ('kernel_stats_pairs192.md', 'kernel_stats_pairs192.md'), ('forward_trace_pairs192.md', 'forward_trace_pairs192.md'),
                      ('host_profile_p1.txt', 'host_profile_pairs1.txt'), (f'{a.tag}_pmc_traffic_kernels.md', 'pmc_traffic_kernels.md')):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, f'{a.tag}_{out}'))
    shutil.copy(os.path.join(src, 'pmc_traffic.json'), os.path.join(dst, 'pmc_traffic.json'))
    with open(os.path.join(src, 'pytest.log')) as f:
        tail = f.read().splitlines()[-14:]
    with open(os.path.join(dst, f'{a.tag}_pytest_tail.txt'), 'w') as f:
        f.write('\n'.join(tail) + '\n')
    with open(os.path.join(dst, f'{a.tag}_e2e_harness.txt'), 'w') as out:
        out.write('# tools/evidence.sh: test.py --benchmark 3DLoMatch --synthetic 1781 (files -> loader -> H2D -> forward -> pose gather -> est.log)\n')
        for name in ('e2e_build', 'e2e_npy', 'e2e_pth', 'e2e_npy_batch64', 'e2e_npy_cold', 'e2e_thread'):
            p = os.path.join(src, name + '.log')
            if os.path.exists(p):
                lines = [l for l in open(p).read().splitlines() if 'End to end' in l or 'loader:' in l or 'materialised' in l]
                out.write(f'## {name}\n' + '\n'.join(lines) + '\n')
    print(sorted(os.path.basename(f) for f in glob.glob(os.path.join(dst, a.tag + '_*'))))


if __name__ == '__main__':
    main()
