"""The reference's OWN demo_free_integration.py, byte for byte, on the drop-in (BASELINE north_star: "drops
into demo_free_integration.py unchanged").  The script is not in this repository and never will be: the test runs when
$GNSS_INS_SIM_REFERENCE names a checkout of Aceinna/gnss-ins-sim (its demo_free_integration.py:19 reads the motion files relative
to the working directory, so the interpreter is started inside the checkout) and is skipped with the reason elsewhere.
tests/test_gpu_sim_dropin.py::test_demo_free_integration_sequence restates the script's calling sequence for boxes without a
checkout."""
import hashlib
import os
import subprocess
import sys

import pytest

from conftest import PKG

pytestmark = pytest.mark.gpu

REF = os.environ.get('GNSS_INS_SIM_REFERENCE', '')
SCRIPT = os.path.join(REF, 'demo_free_integration.py') if REF else ''

RUNNER = r'''
import hashlib, runpy, sys
src = open('demo_free_integration.py', 'rb').read()
print('SCRIPT_SHA256', hashlib.sha256(src).hexdigest())
sys.argv = [sys.argv[1], 'demo_free_integration.py']          # = `python <repo>/gnss-ins-sim_amd/dropin.py demo_free_integration.py`
runpy.run_path(sys.argv[0], run_name='__main__')              # the launcher runs the file as it lies in the checkout
import gnss_ins_sim.sim.ins_sim as s, demo_algorithms.free_integration as f, demo_algorithms.free_integration_odo as o, ginsim
print('MODULES', s.__file__, f.__file__, o.__file__)
print('LIBRARY', ginsim.LIB_PATH)
'''


@pytest.mark.skipif(not (SCRIPT and os.path.isfile(SCRIPT)),
                    reason='$GNSS_INS_SIM_REFERENCE does not name a checkout of the reference (its demo_free_integration.py is never '
                           'copied into this repository)')
def test_the_reference_s_own_demo_script_runs_on_the_drop_in():
    # no PYTHONPATH: the working directory (the checkout, with the reference's own packages in it) is sys.path[0] of `python -c`,
    # exactly the situation of a user in a checkout -- the launcher must put the drop-in in front by itself
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', MPLBACKEND='Agg')
    env.pop('PYTHONPATH', None)
    out = subprocess.run([sys.executable, '-c', RUNNER, os.path.join(PKG, 'dropin.py')], cwd=os.path.abspath(REF), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         universal_newlines=True, timeout=600)
    text = out.stdout
    keep = os.environ.get('GINSIM_TRANSCRIPT')
    if keep:
        with open(keep, 'w') as f:
            f.write('$ cd $GNSS_INS_SIM_REFERENCE && python <repo>/gnss-ins-sim_amd/dropin.py demo_free_integration.py\n')
            f.write(text)
    assert out.returncode == 0, text[-3000:]
    with open(SCRIPT, 'rb') as f:
        assert 'SCRIPT_SHA256 ' + hashlib.sha256(f.read()).hexdigest() in text
    # the packages the script imported are the drop-in's, the arithmetic ran in libginsim.so
    mods = [l for l in text.splitlines() if l.startswith('MODULES ')][0].split()[1:]
    assert all(os.path.abspath(m).startswith(os.path.abspath(PKG)) for m in mods), mods
    assert [l for l in text.splitlines() if l.startswith('LIBRARY ')][0].endswith('libginsim.so')
    # the summary the reference prints (ins_sim.py:339-413) for sim.run(10); results(err_stats_start=-1, gen_kml=True)
    assert 'Simulation runs: 10' in text and 'Simulation time duration: 10.0 s' in text
    assert 'The following are error statistics.' in text
    for section in ('simulation attitude (Euler, ZYX) from algo', 'simulation position from algo', 'simulation velocity from algo'):
        assert '-----------statistics for ' + section in text, section
    for algo in ('algo0', 'algo1'):            # free_integration_odo, free_integration: one end-point group each
        assert text.count('Simulation run %s:' % algo) == 3, algo
    assert text.count('--Max error:') == 6 and text.count('--Avg error:') == 6 and text.count('--Std of error:') == 6
