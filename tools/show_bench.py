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
pl = d.get('placement')
if pl and pl.get('job'):
    a = pl['job'].get('arena', {})
    print('  placement %s: placed %s, arena %s stripes by class %s, search %.2f s (%s chunks created, %s probes, %.0f GiB held at most)' % (
        pl['mode'], pl['job']['placed'], a.get('stripe_classes'), a.get('stripes_of_class'), a.get('search_seconds', 0), a.get('chunks_created'),
        a.get('probes'), a.get('peak_held_bytes', 0) / 2 ** 30))
for k in ('roofline_asis', 'roofline_placed'):
    if k in d and 'frac' in d[k]:
        print('  %s: kernel %.4f ms frac %.3f (the same launch, the other placement, same process)' % (k, d[k]['kernel_ms_avg'], d[k]['frac']))
if 'efficiency' in d:
    print('  N = %d: efficiency %.3f, rccl_ranks %s, per-rank roofline %s' % (d['n_gpus'], d['efficiency'], d.get('rccl_ranks'),
                                                                            ['%.3f' % f for f in d['per_rank']['roofline_frac']]))
print('device', d['config']['device'], 'lib', d['config']['libginsim_sha256'])
for l in d.get('configs', []):
    ro = l.get('roofline')
    if ro:
        print('  %-22s kernel %9.4f ms (min %s)  bound %-4s frac %s  %s' % (l['name'], ro['kernel_ms_avg'], ('%.4f' % l['kernel_ms_min']) if 'kernel_ms_min' in l else '-',
                                                                          ro['bound'], ('%.3f' % ro['frac']) if ro.get('frac') is not None else None,
                                                                          ('hbm %.3f' % ro['hbm']['frac']) if 'hbm' in ro else ''))
    if l.get('placement') and l['placement'].get('unplaced'):
        print('    NOT placed: %s' % l['placement']['unplaced'])
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
