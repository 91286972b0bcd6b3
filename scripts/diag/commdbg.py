"""which RCCL / HIP runtime a process ends up with, by load order (argv[1]: 'ours-first' or 'torch-first'), and whether the
communicator comes up (NCCL_DEBUG=INFO for RCCL's own account)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
order = sys.argv[1] if len(sys.argv) > 1 else "ours-first"
def libs():
    return sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l or "libamdhip" in l))
if order == "torch-first":
    import torch
    torch.zeros(4, device="cuda").sum().item()
from multiclust_amd import hip
import multiclust_amd as mc
lib = hip.load()
ctx = mc.Context(0)
print("after our library:", libs(), flush=True)
if order == "ours-first":
    import torch
    torch.zeros(4, device="cuda").sum().item()
    print("after torch:", libs(), flush=True)
comm = C.c_void_p()
rc = lib.mchip_comm_create(C.byref(comm), 1, (C.c_int * 1)(0))
print("mchip_comm_create rc", rc, libs(), flush=True)
