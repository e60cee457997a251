#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -S listing.

    hipcc ... --cuda-device-only -S -o mc.s csrc/mc_kernel.hip
    python tools/isa_loops.py mc.s <mangled kernel name substring> [min_len]

Finds backward branches (a loop = the lines between a label and a later branch to it) and prints, per loop, the
count of VALU / SALU / VMEM / LDS / SMEM instructions and the most frequent opcodes.  A static count: it says what one
trip through the straight-line body issues, which is what matters for a VALU-issue-bound kernel.
"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and ':' in l and key in l.split(':')[0])
    end = next(i for i in range(start, len(lines)) if lines[i].strip() == 's_endpgm')
    body = lines[start:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r'^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)|^\s+s_branch\s+(\.LBB\d+_\d+)', l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i and i - labels[t] >= min_len:
                loops.append((labels[t], i, t))
    print('%s: %d lines, %d loops >= %d lines' % (body[0].split(':')[0], len(body), len(loops), min_len))
    for a, b, t in loops:
        ops = collections.Counter()
        cls = collections.Counter()
        for l in body[a:b + 1]:
            m = re.match(r'^\s+([a-z_0-9]+)', l)
            if not m:
                continue
            op = m.group(1)
            ops[op] += 1
            if op.startswith('v_'):
                cls['VALU'] += 1
                if 'f64' in op or 'u64' in op or 'i64' in op:
                    cls['VALU64'] += 1
            elif op.startswith(('s_load', 's_buffer')):
                cls['SMEM'] += 1
            elif op.startswith('s_'):
                cls['SALU'] += 1
            elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
                cls['VMEM'] += 1
            elif op.startswith('ds_'):
                cls['LDS'] += 1
        print('\nloop %s: lines %d..%d (%d)  %s' % (t, a, b, b - a, dict(cls)))
        print('   ' + ', '.join('%s %d' % kv for kv in ops.most_common(28)))


if __name__ == '__main__':
    main()
