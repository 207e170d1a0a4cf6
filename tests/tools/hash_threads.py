import os, sys, time
sys.path.insert(0, os.getcwd())
from lens_trace_amd import synth
from lens_trace_amd.renderer import RendererHIP
s = synth.heightfield_wall(708).validate()
r = RendererHIP(0)
r.set_scene(s)
for th in ("2", "4", "8", "12", "16", "24", "32"):
    os.environ["LT_HOST_THREADS"] = th
    ts = []
    for k in range(12):
        t = time.perf_counter(); r.set_scene(s); ts.append((time.perf_counter() - t) * 1e3)
    print("LT_HOST_THREADS=%s: unchanged set_scene (hash only) min %.3f median %.3f ms" % (th, min(ts), sorted(ts)[len(ts)//2]))
