"""Which HIP/HSA/RCCL copies end up in the process, by import order (torch bundles its own)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
order = sys.argv[1]
if order == "torch_first":
    import torch
from metropolismontecarlo_amd import _lib
L = _lib.lib()
if order == "lib_first":
    import torch
def maps():
    seen = set()
    for l in open("/proc/self/maps"):
        p = l.split()[-1]
        if any(k in p for k in ("amdhip", "hsa-runtime", "rccl")) and p not in seen:
            seen.add(p); print("   ", p)
maps()
ident = C.create_string_buffer(128)
print("unique_id", L.mmc_dist_unique_id(ident))
d = C.c_void_p()
rc = L.mmc_dist_init(0, 1, ident, 0, C.byref(d))
print("init", rc, L.mmc_last_error())
maps()
