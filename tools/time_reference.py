#!/usr/bin/env python3
"""Time the UNMODIFIED reference's own CPU path (Sim.run on BASELINE config 1: 90-degree turn @100 Hz, 'mid-accuracy'
6-axis IMU, ref_frame 1, FreeIntegration; the reference is single-threaded) and write the host-stamped record
profiles/reference_cpu.json.  Runs only where /root/reference exists (the build container); bench.py quotes the record on the
GPU box, where the reference cannot be imported.

    python tools/time_reference.py [runs]"""
import datetime
import json
import os
import platform
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or 'unknown'


def main():
    import bench
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    rec = bench.reference_python_baseline(runs * 1000 / 4.76e4)
    if rec.get('kind') != 'reference' or not rec.get('value'):
        sys.exit('the reference is not importable here: %r' % (rec,))
    try:
        head = subprocess.run(['git', '-C', '/root/reference', 'rev-parse', 'HEAD'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                              universal_newlines=True).stdout.strip()
    except OSError:
        head = ''
    out = {'value': rec['value'], 'unit': rec['unit'], 'cores': 1, 'sample': rec['sample'], 'host': platform.node(), 'cpu': cpu_model(),
           'logical_cpus': os.cpu_count(), 'date': datetime.datetime.now(datetime.timezone.utc).strftime('%Y-%m-%dT%H:%MZ'),
           'python': platform.python_version(), 'numpy': __import__('numpy').__version__, 'reference_head': head,
           'made_by': 'tools/time_reference.py'}
    path = os.path.join(REPO, 'profiles', 'reference_cpu.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write('\n')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
