"""The bench's Allan leg alone (development aid): GINSIM_LIB selects the library build."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import bench
import ginsim
from ginsim import workloads
ctx = ginsim.Context(0)
leg = bench.leg_allan(ginsim, workloads, ctx)
print(json.dumps({k: leg[k] for k in leg if k in ('name', 'e2e_wall_ms', 'gen_ms')} | {'kernel_ms_avg': leg['roofline']['kernel_ms_avg'], 'frac': leg['roofline']['frac']}))
