#!/usr/bin/env python3
"""Per-kernel registers / scratch / occupancy as hipcc reported them at the last build (build/<file>.resources.txt),
with demangled names.  `tools/resources.py [file-stem ...]` prints a table; `--csv PATH` writes it (profiles/<tag>_kernel_resources.csv)."""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(REPO, 'gnss-ins-sim_amd', 'build')


def demangle(names):
    for exe in ('c++filt', '/opt/rocm/llvm/bin/llvm-cxxfilt'):
        try:
            out = subprocess.run([exe], input='\n'.join(names), stdout=subprocess.PIPE, universal_newlines=True, check=True).stdout
            return out.strip().split('\n')
        except (OSError, subprocess.CalledProcessError):
            continue
    return names


def table(stems=None):
    rows = []
    for f in sorted(os.listdir(BUILD)):
        if not f.endswith('.resources.txt') or (stems and f.split('.')[0] not in stems):
            continue
        txt = open(os.path.join(BUILD, f)).read()
        blocks = re.split(r'Function Name: ', txt)[1:]
        names = demangle([b.split('\n')[0].strip() for b in blocks])
        for name, b in zip(names, blocks):
            def g(k):
                m = re.search(k + r': (\S+)', b)
                return m.group(1) if m else ''
            rows.append((f.split('.')[0], re.sub(r'\(.*$', '', name.replace('void ', '')), g('VGPRs'), g('AGPRs'), g('SGPRs'),
                         g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
    return rows


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    csv = None
    if '--csv' in sys.argv:
        csv = sys.argv[sys.argv.index('--csv') + 1]
        args = [a for a in args if a != csv]
    rows = table(args or None)
    if csv:
        with open(csv, 'w') as f:
            f.write('file,kernel,vgprs,agprs,sgprs,scratch_bytes_per_lane,occupancy_waves_per_simd,static_lds_bytes\n')
            for r in rows:
                f.write(','.join('"%s"' % x if ',' in x else x for x in r) + '\n')
    else:
        for r in rows:
            print('%-14s %-72s V %3s A %2s S %3s scr %4s occ %s lds %s' % r)
