import sys, os, time
sys.path[:0] = ['/root/repo/gnss-ins-sim_amd', '/root/repo']
import ginsim
ctx = ginsim.Context(0)
G = 1 << 30
def free(): return ctx.mem_info()[0] / G
print('start free %.2f GiB' % free())
for k in range(4):
    f0 = free()
    ok = ctx.placed_reserve(3 * G)
    i = ctx.placed_info()
    f1 = free()
    b = ctx.malloc(2 * G, placed=True)
    b.free()
    ctx.release_pool()
    time.sleep(0.3)
    f2 = free()
    print('cycle %d: reserve ok=%s mapped %.1f GiB created %d searches %d | free before %.2f, with arena %.2f, after release %.2f' % (k, ok, i['mapped_bytes'] / G, i['chunks_created'], i['searches'], f0, f1, f2))
ctx.close()
