"""N > 1 on real hardware.  The GPU box has ONE device, so two ranks share GPU 0 and exchange over gloo; the code
path (sharding by global run id, per-rank launch, one all-reduce of the statistics record, merge) is the one the
8-GPU RCCL run takes."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, PKG

pytestmark = pytest.mark.gpu


def _port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(nproc, extra, launcher=True):
    env = dict(os.environ, OMP_NUM_THREADS='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr',
           '127.0.0.1', '--master-port', str(_port()), os.path.join(REPO, 'bench.py'), '--gpus', str(nproc)] + extra
    if nproc == 1:
        cmd = [sys.executable, os.path.join(REPO, 'bench.py')] + extra
    elif not launcher:          # the driver's BENCH form of the command: no launcher in front
        cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(nproc)] + extra
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    line = [l for l in out.stdout.decode().splitlines() if l.startswith('{')][-1]
    return json.loads(line)


def test_bench_two_ranks_share_the_work():
    common = ['--steps', '2', '--warmup', '1', '--runs-per-gpu', '8192', '--cpu-baseline-seconds', '0', '--no-legs', '--pmc', 'off']
    one = _bench(1, common)
    two = _bench(2, common + ['--backend', 'gloo', '--shared-device'])
    assert two['n_gpus'] == 2 and two['scaling'] == 'weak' and two['config']['total_runs_per_step'] == 16384
    assert two['result']['runs'] == 16384 and one['result']['runs'] == 8192
    # merged statistics of 16 384 runs vs 8 192 runs of the same distribution
    np.testing.assert_allclose(two['result']['att_std_deg'], one['result']['att_std_deg'], rtol=0.05)
    assert two['value'] > 0 and 'roofline' in two and 'cpu_baseline' not in two and 'N = 1 only' in two['cpu_baseline_note']


def test_bench_starts_itself_for_n_gt_1():
    """`python bench.py --gpus 2 ...` with NO launcher in front (the form of the driver's command): the script re-executes itself
    under torch.distributed.run with two ranks; the N > 1 line carries per_gpu_single (rank 0 alone on the same per-GPU load)
    and the time-based pre-warm."""
    common = ['--steps', '3', '--warmup', '1', '--runs-per-gpu', '8192', '--cpu-baseline-seconds', '0', '--no-legs', '--pmc', 'off']
    two = _bench(2, common + ['--backend', 'gloo', '--shared-device'], launcher=False)
    assert two['n_gpus'] == 2 and two['config']['total_runs_per_step'] == 16384 and two['result']['runs'] == 16384
    s = two['per_gpu_single']
    assert s['runs'] == 8192 and s['value'] > 0 and s['unit'] == 'sample*MC/s'
    assert two['prewarm_ms'] >= 30.0 and 'gloo' in two['config']['parallelism']
    # every rank is in the line: its kernel time in the timed region and its own single-GPU reference
    assert len(two['per_rank']['kernel_ms_avg']) == 2 and two['per_rank']['max'] >= two['per_rank']['min'] > 0
    assert two['per_rank']['argmax'] in (0, 1) and len(s['every_rank']['value']) == 2 and s['every_rank']['min'] > 0
    # two ranks sharing ONE GPU cannot beat one rank by more than the overlap of host work: a sanity bound, not a scaling claim
    assert two['value'] < 2.5 * s['value']


def test_bench_eight_ranks_without_a_launcher():
    """The driver's 8-GPU command as it will be issued -- `python bench.py --gpus 8`, no launcher -- with the eight ranks sharing
    the one GPU of this box over gloo: self-start under torch.distributed.run, sharding by global run id, the per-rank turns of
    per_gpu_single, the exchange, the merge and the per-rank fields of the line all execute with world = 8."""
    d = _bench(8, ['--steps', '2', '--warmup', '1', '--runs-per-gpu', '4096', '--cpu-baseline-seconds', '0', '--no-legs', '--pmc', 'off',
                   '--backend', 'gloo', '--shared-device'], launcher=False)
    assert d['n_gpus'] == 8 and d['config']['total_runs_per_step'] == 8 * 4096 and d['result']['runs'] == 8 * 4096
    assert len(d['per_rank']['kernel_ms_avg']) == 8 and 0 <= d['per_rank']['argmax'] < 8
    assert len(d['per_gpu_single']['every_rank']['value']) == 8 and d['per_gpu_single']['runs'] == 4096
    assert d['scaling'] == 'weak' and d['value'] > 0


def test_bench_line_contract():
    """The ONE JSON line the driver reads: every field of the contract, on the C2 workload at its real size (few steps), with
    a short CPU-baseline sample and one leg-free pass; the roofline object must be consistent with itself."""
    d = _bench(1, ['--steps', '5', '--warmup', '3', '--cpu-baseline-seconds', '1', '--no-legs', '--pmc', 'off'])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 5 and d['warmup'] == 3 and d['higher_is_better'] is True and d['vs_baseline'] is None
    assert d['dtype'] == 'f64' and d['data'] == 'synthetic' and d['scaling'] == 'weak' and d['unit'] == 'sample*MC/s'
    assert 'workload' in d['config'] and 'configs[1]' in d['config']['workload'] and 'model' not in d['config']
    assert d['config']['runs_per_gpu'] == 65536 and d['config']['samples_per_run'] == 1000
    assert 'no collective' in d['config']['parallelism'] and d['prewarm_ms'] >= 30.0 and 'per_gpu_single' not in d
    # value = units of all timed steps / wall time
    np.testing.assert_allclose(d['value'], 65536 * 1000 / (d['ms_per_step'] * 1e-3), rtol=1e-6)
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and r['traffic'] is None      # --pmc off
    assert r['algorithmic_bytes_per_launch'] == 120 * 65536 * 1000 + 72 * 65536
    np.testing.assert_allclose(r['achieved'], r['algorithmic_bytes_per_launch'] / (r['kernel_ms_avg'] * 1e-3) / 1e9, rtol=1e-9)
    np.testing.assert_allclose(r['frac'], r['achieved'] / r['peak'], rtol=1e-12)
    assert 0.3 < r['frac'] < 0.95 and r['kernel_ms_avg'] < d['ms_per_step'] and 'mc_kernel' in r['kernel']
    # the same K steps once more, a few seconds later, in the same process: same statistics, its own clock
    h = d['headline_again']
    assert h['same_statistics'] is True and h['seconds_after_the_first'] >= 3.0 and 0.5 < h['value'] / d['value'] < 2.0
    np.testing.assert_allclose(r['frac_again'], r['algorithmic_bytes_per_launch'] / (h['kernel_ms_avg'] * 1e-3) / 1e9 / r['peak'], rtol=1e-9)
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['unit'] == 'sample*MC/s' and c['cores'] >= 1 and c['value'] > 0 and 'sample' in c
    assert c['reference_python']['kind'] in ('quoted', 'reference') and c['reference_python']['cores'] == 1


def test_rccl_exchange_executes_on_one_gpu():
    """The box has one GPU, so the 8-rank RCCL run cannot happen here -- but its code path can: backend 'nccl' (= RCCL) with a
    one-rank communicator executes communicator set-up, the device-tensor all-reduce of the statistics table, the
    non-blocking work handles and the barrier exactly as N ranks would; the merged record must equal the local one."""
    common = ['--steps', '3', '--warmup', '1', '--runs-per-gpu', '8192', '--cpu-baseline-seconds', '0', '--no-legs', '--pmc', 'off']
    plain = _bench(1, common)
    rccl = _bench(1, common + ['--force-dist', '--backend', 'nccl', '--exchange', 'torch'])
    assert rccl['n_gpus'] == 1 and 'all-reduce (RCCL)' in rccl['config']['parallelism']
    assert rccl['result'] == plain['result']
    # the default with backend nccl: the library's own communicator, the all-gather on the kernel stream (C ABI), after one
    # batch was cross-checked against the torch.distributed exchange
    abi = _bench(1, common + ['--force-dist', '--backend', 'nccl'])
    assert 'behind the C ABI' in abi['config']['parallelism'] and 'unavailable' not in abi['config']['parallelism'] \
        and 'disagreed' not in abi['config']['parallelism'], abi['config']['parallelism']
    assert abi['result'] == plain['result']
    # the line's proof that RCCL saw the ranks: read back from the communicator itself (ncclCommCount / UserRank / CuDevice)
    assert abi['rccl_ranks'] == 1 and abi['every_rank'] == [{'ranks': 1, 'rank': 0, 'device': 0}], abi.get('every_rank')
    assert 'rccl_ranks' not in plain


def test_arena_search_then_collectives_with_two_ranks():
    """the N > 1 control flow WITH the placement -- every rank builds its placed arena (the chunk search, on
    the one GPU they share, each holding a capped share of it), then the ranks meet in the collectives: per_gpu_single turns,
    warm-up, timed steps, exchange.  16 384 runs per rank materialise 1.97 GB: placed by default.  The line carries every rank's
    placement and roofline and its own efficiency figure."""
    d = _bench(2, ['--steps', '4', '--warmup', '2', '--runs-per-gpu', '16384', '--cpu-baseline-seconds', '0', '--no-legs', '--pmc', 'off',
                   '--backend', 'gloo', '--shared-device'], launcher=False)
    assert d['n_gpus'] == 2 and d['result']['runs'] == 2 * 16384
    pl = d['per_rank']['placement']
    assert len(pl) == 2 and all(p['placed'] == ['imu', 'traj_free'] and sum(p['arena_stripes_of_class']) >= 4 for p in pl), pl
    assert d['placement']['mode'] == 'placed' and d['placement']['job']['arena']['classes'] >= 2
    assert len(d['per_rank']['roofline_frac']) == 2 and all(0.05 < f < 0.95 for f in d['per_rank']['roofline_frac'])
    np.testing.assert_allclose(d['efficiency'], d['value'] / (2 * d['per_gpu_single']['every_rank']['min']), rtol=1e-12)
    assert 'rccl_ranks' not in d                      # gloo: no communicator of the library's own to ask


def test_abi_exchange_one_rank_communicator():
    """ginsim_comm_* / ginsim_end_stats_all_*: a one-rank RCCL communicator created by the library itself (librccl resolved at
    run time), the all-gather enqueued on the context's stream behind the reduction; the merged record is the local one, two
    slots can be in flight, and a second communicator on the same context is refused."""
    import ginsim
    from ginsim import workloads
    ctx = ginsim.Context(0)
    ctx.comm_init(1, 0, ctx.comm_unique_id())
    with pytest.raises(ValueError, match='already has a communicator'):
        ctx.comm_init(1, 0, ctx.comm_unique_id())
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    job = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=5000, seed=3)
    job.launch()
    job.stats_all_begin('free', 0)
    job.params.run_offset = 5000
    job.launch()
    job.stats_all_begin('free', 1)
    first, second = job.stats_all_finish(0), job.stats_all_finish(1)
    local = job.stats('free')                        # the second batch is the one still in the buffers
    assert second.count == 5000 and np.array_equal(second.mean, local.mean) and np.array_equal(second.m2, local.m2)
    assert first.count == 5000 and not np.array_equal(first.mean, second.mean)
    with pytest.raises(ValueError, match='nothing was begun'):
        job.stats_all_finish(0)
    ctx.comm_destroy()
    with pytest.raises(ValueError, match='no communicator'):
        job.stats_all_begin('free', 0)
    job.release()
    ctx.close()


_SIM_WORKER = r'''
import os, sys
sys.path[:0] = [%(pkg)r, %(repo)r]
import numpy as np, torch.distributed as dist
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=int(sys.argv[1]), world_size=2)
os.environ['LOCAL_RANK'] = '0'
from gnss_ins_sim.sim import imu_model, ins_sim
from demo_algorithms import free_integration
csv = os.path.join(%(pkg)r, 'motion_profiles', 'turn_90deg.csv')
ini = np.genfromtxt(csv, delimiter=',', skip_header=1, max_rows=1)
ini[0:2] *= np.pi / 180; ini[6:9] *= np.pi / 180
imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu, algorithm=free_integration.FreeIntegration(ini), seed=77)
sim.run(1001)
sim.results(err_stats_start=-1)
keys = list(sim.dmgr.accel.data.keys())
np.save(sys.argv[2], np.concatenate([sim.err_stats['vel']['std'], sim.err_stats['att_euler']['max'], [keys[0], keys[-1], len(keys)]]))
dist.barrier(); dist.destroy_process_group()
'''


_SIM_NCCL_WORKER = r'''
import os, sys, json
sys.path[:0] = [%(pkg)r, %(repo)r]
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%(port)d', rank=0, world_size=1, device_id=torch.device('cuda', 0))
os.environ['LOCAL_RANK'] = '0'
from gnss_ins_sim.sim import imu_model, ins_sim
from demo_algorithms import free_integration
csv = os.path.join(%(pkg)r, 'motion_profiles', 'turn_90deg.csv')
ini = np.genfromtxt(csv, delimiter=',', skip_header=1, max_rows=1)
ini[0:2] *= np.pi / 180; ini[6:9] *= np.pi / 180
imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
out = {}
for rep in range(2):            # the second Sim finds the communicator on the process-wide context
    sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu, algorithm=free_integration.FreeIntegration(ini), seed=77)
    sim.run(777)
    sim.results(err_stats_start=-1)
    out['exchange%%d' %% rep] = sim.mc.exchange
    out['vel_std'] = [float(x) for x in sim.err_stats['vel']['std']]
    out['att_max'] = [float(x) for x in sim.err_stats['att_euler']['max']]
print('RESULT ' + json.dumps(out))
dist.barrier(); dist.destroy_process_group()
'''


def test_sim_uses_the_abi_exchange_with_backend_nccl(tmp_path):
    """Sim under torch.distributed with backend nccl (= RCCL): the end-point records are merged by the library's own all-gather
    behind the C ABI (ginsim_comm_* / ginsim_end_stats_all_*: the exchange bench.py times), after its first result was
    cross-checked against the torch.distributed all-reduce -- here with the one-rank communicator this box allows; the statistics
    equal those of the plain single-process Sim."""
    script = tmp_path / 'n.py'
    script.write_text(_SIM_NCCL_WORKER % {'pkg': PKG, 'repo': REPO, 'port': _port()})
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=env)
    text = p.stdout.decode()
    assert p.returncode == 0, text[-3000:]
    got = json.loads([l for l in text.splitlines() if l.startswith('RESULT ')][-1][7:])
    assert got['exchange0'] == 'abi' and got['exchange1'] == 'abi', got
    sys.path[:0] = [PKG]
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    ini = np.genfromtxt(csv, delimiter=',', skip_header=1, max_rows=1)
    ini[0:2] *= np.pi / 180
    ini[6:9] *= np.pi / 180
    sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False),
                      algorithm=free_integration.FreeIntegration(ini), seed=77)
    sim.run(777)
    sim.results(err_stats_start=-1)
    np.testing.assert_allclose(got['vel_std'], sim.err_stats['vel']['std'], rtol=1e-12)
    np.testing.assert_array_equal(got['att_max'], sim.err_stats['att_euler']['max'])


def test_sim_shards_runs_across_ranks(tmp_path):
    """Sim under torch.distributed (2 ranks): each rank holds its shard, statistics are those of all 1001 runs."""
    script = tmp_path / 'w.py'
    script.write_text(_SIM_WORKER % {'pkg': PKG, 'repo': REPO, 'port': _port()})
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(tmp_path / ('r%d.npy' % r))], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    a, b = np.load(tmp_path / 'r0.npy'), np.load(tmp_path / 'r1.npy')
    np.testing.assert_array_equal(a[:6], b[:6])                 # every rank ends with the same merged statistics
    assert (a[6], a[7], a[8]) == (0, 500, 501) and (b[6], b[7], b[8]) == (501, 1000, 500)
    # reference: one process, all runs
    sys.path[:0] = [PKG]
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    ini = np.genfromtxt(csv, delimiter=',', skip_header=1, max_rows=1)
    ini[0:2] *= np.pi / 180
    ini[6:9] *= np.pi / 180
    sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False),
                      algorithm=free_integration.FreeIntegration(ini), seed=77)
    sim.run(1001)
    sim.results(err_stats_start=-1)
    np.testing.assert_allclose(a[:3], sim.err_stats['vel']['std'], rtol=1e-10)
    np.testing.assert_array_equal(a[3:6], sim.err_stats['att_euler']['max'])


def test_more_ranks_than_runs(tmp_path):
    """Sim.run(1) under 2 ranks -- rank 1 holds no runs, contributes an empty record to the all-reduce and
    must neither crash nor leave rank 0 waiting in the collective."""
    script = tmp_path / 'w1.py'
    script.write_text((_SIM_WORKER % {'pkg': PKG, 'repo': REPO, 'port': _port()}).replace('sim.run(1001)', 'sim.run(1)').replace(
        "keys = list(sim.dmgr.accel.data.keys())", "keys = list(sim.dmgr.accel.data.keys()) if 'accel' in sim.dmgr.available else []").replace(
        "[keys[0], keys[-1], len(keys)]", "[len(keys)]"))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(tmp_path / ('s%d.npy' % r))], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    a, b = np.load(tmp_path / 's0.npy'), np.load(tmp_path / 's1.npy')
    np.testing.assert_array_equal(a[:6], b[:6])
    assert np.all(a[:3] == 0.0) and np.all(a[3:6] > 0.0)       # one run: std 0, max |e| > 0
    assert (a[6], b[6]) == (1, 0)


@pytest.mark.parametrize('keep', ['True', 'False'])
def test_more_ranks_than_runs_with_the_ned_record(tmp_path, keep):
    """Sim.run(1) under 2 ranks in ref_frame 0 with extra_opt='ned'.  Rank 1 holds no runs; which collective merges
    the NED record (recomputed from kept trajectories, or the kernel's own NED end-point record of a statistics-only launch) is
    decided from the Sim configuration -- the same on both ranks -- so neither rank is left alone in a collective."""
    script = tmp_path / 'wn.py'
    script.write_text((_SIM_WORKER % {'pkg': PKG, 'repo': REPO, 'port': _port()})
                      .replace('ref_frame=1', 'ref_frame=0, keep_trajectories=%s' % keep)
                      .replace('sim.run(1001)', 'sim.run(1)')
                      .replace("sim.results(err_stats_start=-1)", "sim.results(err_stats_start=-1, extra_opt='ned')")
                      .replace("sim.err_stats['vel']['std']", "sim.err_stats['pos']['max']")
                      .replace("keys = list(sim.dmgr.accel.data.keys())", "keys = [0]")
                      .replace("[keys[0], keys[-1], len(keys)]", "[len(keys)]"))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(tmp_path / ('n%d.npy' % r))], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    a, b = np.load(tmp_path / 'n0.npy'), np.load(tmp_path / 'n1.npy')
    np.testing.assert_array_equal(a[:6], b[:6])
    assert np.all(a[:3] > 0.0) and np.all(a[:3] < 10.0)        # NED metres after 10 s of a mid-accuracy IMU, not radians
