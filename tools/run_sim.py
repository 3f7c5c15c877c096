"""Developer tool: a few RK steps of one workload through the C++ host layer exactly as bench.py's legs run them (region timers off:
the energy solve beside the velocity solve), for profiling.  usage: python tools/run_sim.py <warmup> <steps> <laghos options...>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laghos_amd import host_lib
warm, steps = int(sys.argv[1]), int(sys.argv[2])
sim = host_lib.Sim(sys.argv[3:] + ["-pa", "-tf", "1e9", "-ms", "1000000", "-vs", "1000000000", "-q"])
sim.enable_timers(False)
for _ in range(warm):
    sim.step()
sim.sync()
t0 = time.perf_counter()
for _ in range(steps):
    sim.step()
sim.sync()
print("ms per step", 1e3 * (time.perf_counter() - t0) / steps, "|e|", sim.e_norm())
sim.close()
