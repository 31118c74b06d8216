"""Which ROCm runtime libraries are mapped in this process (rt_runtime_libraries), and does a context come up? Run it plainly and under rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import rta
p = rta.load()
libs, ok = p.runtime_libraries()
print("one runtime:", ok)
for l in libs: print("  ", l)
c = p.Context(0)
print("context ok")
