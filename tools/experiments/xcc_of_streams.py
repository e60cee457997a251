import sys
sys.path.insert(0, 'gnss-ins-sim_amd')
import ginsim
c = ginsim.default_context()
print('main', c.first_xcc(), c.first_xcc())
cs = []
for i in range(10):
    s = ginsim.Context(0)
    cs.append(s)
    print('side', i, s.first_xcc(), s.first_xcc(), 'main again', c.first_xcc())
for s in cs[:5]:
    s.close()
for i in range(4):
    s = ginsim.Context(0)
    print('after closing five: side', s.first_xcc())
    cs.append(s)
