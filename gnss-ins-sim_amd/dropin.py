#!/usr/bin/env python3
"""Run a script of the reference on the drop-in packages, the script itself unchanged:

    cd /path/to/gnss-ins-sim                                  # the reference checkout (its demos read ./demo_motion_def_files)
    python /path/to/repo/gnss-ins-sim_amd/dropin.py demo_free_integration.py [script arguments]

Why a launcher: `python demo_free_integration.py` puts the SCRIPT's directory first on sys.path, and in a checkout that directory
holds the reference's own gnss_ins_sim/ and demo_algorithms/ -- they would win over anything on $PYTHONPATH.  Started through this
file, sys.path[0] is this directory, so `from gnss_ins_sim.sim import ins_sim` (demo_free_integration.py:13-14) and
`from demo_algorithms import free_integration` (:44-45) resolve to the drop-in; everything the drop-in does not rebuild falls
through to the checkout ($GNSS_INS_SIM_REFERENCE, set here to the script's directory when it is a checkout and the variable is
unset; gnss_ins_sim/_reference.py)."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    script = sys.argv[1]
    if sys.path[0] != HERE:
        sys.path.insert(0, HERE)
    where = os.path.dirname(os.path.abspath(script))
    if 'GNSS_INS_SIM_REFERENCE' not in os.environ and os.path.isfile(os.path.join(where, 'gnss_ins_sim', 'sim', 'ins_sim.py')):
        os.environ['GNSS_INS_SIM_REFERENCE'] = where
    sys.argv = sys.argv[1:]
    runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
    main()
