#!/usr/bin/env python3
"""Build libginsim.so (HIP kernels + C ABI) for gfx950, in-tree.

    python gnss-ins-sim_amd/build.py [--force] [--verbose]

hipcc cross-compiles without a GPU.  Objects go to gnss-ins-sim_amd/build/, the library to
gnss-ins-sim_amd/lib/libginsim.so (git-ignored, but shipped to the GPU box by gpurun).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'lib', 'libginsim.so')
ARCH = 'gfx950'
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

COMMON = ['-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(REPO, 'include'), '-I' + CSRC, '-Wall',
          '-Wno-unused-function']
# (source, extra flags)
SOURCES = [
    # machine-LICM hoists ~35 fp64 polynomial constants (SGPR pairs) out of the time loop and then spills them
    # to VGPR lanes; without it they are re-materialised with s_mov next to their use (SGPR spills 192 -> 71)
    # -ffp-contract=on: fused multiply-adds only where one source expression spells a*b+c, decided in the front
    # end, so the plain and the wave-specialised kernels (same inlined functions) give identical bits
    ('mc_kernel.hip', ['--offload-arch=' + ARCH, '-mllvm', '-disable-machine-licm', '-ffp-contract=on']),
    # the fp32 kernel is DEFINED operation by operation (the float oracle repeats it to the bit): no contraction at all,
    # fused multiply-adds only where the source spells __builtin_fmaf
    # no SLP vectorisation either: it packed the consumer's float arithmetic into v_pk_*_f32 pairs (a 4-cycle-class
    # instruction) and paid for the pairing with v_mov: 53 packed + 38 moves of 152 VALU on the common path;
    # scalar, the same step is 169 plain VALU and the launch 15 % faster (nothing kept 0.685 -> 0.584 ms), 132 -> 120 VGPRs
    ('mc_kernel_f32.hip', ['--offload-arch=' + ARCH, '-mllvm', '-disable-machine-licm', '-ffp-contract=off', '-fno-slp-vectorize']),
    ('stats.hip', ['--offload-arch=' + ARCH]),
    ('allan.hip', ['--offload-arch=' + ARCH]),
    ('placed.hip', ['--offload-arch=' + ARCH]),
    ('vib_psd.hip', ['--offload-arch=' + ARCH]),
    ('ginsim_api.hip', ['--offload-arch=' + ARCH]),
    # host-only truth generator; no fused multiply-adds (see the file header)
    ('pathgen.cpp', ['-x', 'c++', '-ffp-contract=off']),
    # RCCL behind the C ABI (host code; librccl is dlopen()ed at run time, nothing is linked)
    ('comm.cpp', ['-x', 'hip', '--offload-arch=' + ARCH]),
]


def newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False, tag=None, defines=(), xflags=()):
    """tag / defines: an experiment build (A/B timing) -> lib/libginsim_<tag>.so from build/<tag>/ objects compiled with
    -D<define>; loaded with GINSIM_LIB=<path>.  The product build has neither."""
    global OBJ, LIB
    if tag:
        OBJ = os.path.join(HERE, 'build', tag)
        LIB = os.path.join(HERE, 'lib', 'libginsim_%s.so' % tag)
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hpp', '.h'))]
    deps.append(os.path.join(REPO, 'include', 'ginsim.h'))
    deps.append(os.path.abspath(__file__))
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(OBJ, src.rsplit('.', 1)[0] + '.o')
        objs.append(o)
        if force or newer(s, o) or any(newer(d, o) for d in deps):
            cmd = [HIPCC] + COMMON + extra + ['-D' + d for d in defines] + (list(xflags) if src.startswith('mc_kernel') else []) + ['-c', s, '-o', o]
            if src.endswith('.hip'):
                # per-kernel registers / scratch / occupancy as the compiler reports them -> build/<name>.resources.txt
                # (tests/test_host_cpu.py holds the hot kernels to two wavefronts per SIMD and no AGPR spills)
                cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
            if verbose:
                print(' '.join(cmd))
            p = subprocess.run(cmd, stderr=subprocess.PIPE, universal_newlines=True)
            if p.returncode != 0:       # show the compiler's own message, not a parse error of its remarks
                sys.stderr.write('\n'.join(l for l in p.stderr.splitlines() if 'remark:' not in l) + '\n')
                raise subprocess.CalledProcessError(p.returncode, cmd)
            if src.endswith('.hip'):
                remarks = [l for l in p.stderr.splitlines() if 'kernel-resource-usage' in l and 'remark:' in l]
                with open(o[:-2] + '.resources.txt', 'w') as f:
                    f.write('\n'.join(l.split('remark:', 1)[1].replace('[-Rpass-analysis=kernel-resource-usage]', '').strip()
                                      for l in remarks) + '\n')
                other = [l for l in p.stderr.splitlines() if 'kernel-resource-usage' not in l and not l.startswith(' ')]
                other = [l for l in other if l.strip() and not l.lstrip().startswith('|')]
                if p.returncode != 0 or verbose:
                    sys.stderr.write(p.stderr if p.returncode != 0 else '\n'.join(other) + '\n')
            elif p.stderr:
                sys.stderr.write(p.stderr)
            if p.returncode != 0:
                raise subprocess.CalledProcessError(p.returncode, cmd)
    if force or any(newer(o, LIB) for o in objs):
        cmd = [HIPCC, '-shared', '-fPIC', '--offload-arch=' + ARCH, '-o', LIB] + objs + ['-ldl']
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    _tag = sys.argv[sys.argv.index('--tag') + 1] if '--tag' in sys.argv else None
    _defs = [a[2:] for a in sys.argv if a.startswith('-D')]
    _xf = ['-' + a[2:] for a in sys.argv if a.startswith('-X')]      # -X<flag>: extra compiler flag for the MC kernels (experiments)
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv or '-v' in sys.argv, tag=_tag, defines=_defs, xflags=_xf))
