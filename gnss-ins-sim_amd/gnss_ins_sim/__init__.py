"""Drop-in ``gnss_ins_sim`` package whose Monte-Carlo hot path runs on MI355X through libginsim.so.

Put ``gnss-ins-sim_amd/`` on ``sys.path`` (ahead of the reference checkout, if any) and scripts such as the
reference's ``demo_free_integration.py`` run unchanged:

    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration

Only the path of SURVEY.md section 8 is provided (Sim / IMU / FreeIntegration plugins / error statistics);
plotting, KML export, the GUI bridge and the other demo algorithms are out of scope.
"""
from . import _reference as _reference          # noqa: E402

_reference.install()        # modules this package lacks come from $GNSS_INS_SIM_REFERENCE when it names a checkout
