#!/usr/bin/env python3
"""Print the figures of a bench.py JSON line (file argument) the way DESIGN.md section 6 quotes them."""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d['roofline']
print('headline  %.4g %s  %.4f ms/step  kernel %.4f ms  frac %.3f  traffic %s' % (d['value'], d['unit'], d['ms_per_step'], r['kernel_ms_avg'], r['frac'], r.get('traffic')))
if 'headline_again' in d:
    h = d['headline_again']
    print('  again   %.4g  %.4f ms/step  kernel %.4f ms  frac %.3f  (%.1f s later)' % (h['value'], h['ms_per_step'], h['kernel_ms_avg'], r.get('frac_again', 0), h['seconds_after_the_first']))
pl = d['config'].get('placement')
if pl:
    print('  placement: moved %s after %s candidate(s): launch %s -> %s ms %s' % (pl.get('moved'), pl.get('candidates'), pl.get('launch_ms_before', pl.get('launch_ms')),
                                                                             pl.get('launch_ms'), pl.get('why', '')))
print('device', d['config']['device'], 'lib', d['config']['libginsim_sha256'])
for l in d.get('configs', []):
    ro = l.get('roofline')
    if ro:
        print('  %-22s kernel %9.4f ms (min %s)  bound %-4s frac %s  %s' % (l['name'], ro['kernel_ms_avg'], ('%.4f' % l['kernel_ms_min']) if 'kernel_ms_min' in l else '-',
                                                                          ro['bound'], ('%.3f' % ro['frac']) if ro.get('frac') is not None else None,
                                                                          ('hbm %.3f' % ro['hbm']['frac']) if 'hbm' in ro else ''))
    if l.get('placement'):
        q = l['placement']
        print('    placement: moved %s after %s candidate(s), %s -> %s ms %s' % (q.get('moved'), q.get('candidates'), q.get('launch_ms_before', q.get('launch_ms')), q.get('launch_ms'), q.get('why', '')))
    if l['name'] == 'C5_allan_end_to_end':
        g = l['sensor_generation_roofline']
        print('    generation %.4f ms (min %.4f) bound %s frac %.3f %s; allan wall %.4f ms, call min %.4f ms, traffic/alg %s' % (
            l['sensor_generation_ms'], l['sensor_generation_ms_min'], g['bound'], g['frac'], ('hbm %.3f' % g['hbm']['frac']) if 'hbm' in g else '',
            l['relayout_plus_allan_wall_ms'], l['allan_call_ms_min'], ro.get('traffic_over_algorithmic')))
        if 'same_series_in_another_allocation' in l:
            o = l['same_series_in_another_allocation']
            print('    another allocation of the same series: %.4f ms avg (min %.4f) frac %.3f' % (o['kernel_ms_avg'], o['ms_min'], o['frac']))
    if l['name'] == 'sim_e2e':
        for t in ('C2', 'C3'):
            print('    sim %s run %.4f s results %.4f s  %.4g sample*MC/s  walls %s %s' % (t, l[t]['run_wall_s'], l[t]['results_wall_s'], l[t]['sample_MC_per_s_end_to_end'],
                                                                                    ['%.3f' % w for w in l[t]['wall_s_every_construction']], l[t].get('imu', '')))
c = d.get('cpu_baseline')
if c:
    print('cpu %.4g %s on %d cores (%s)' % (c['value'], c['unit'], c['cores'], c['kind']))
