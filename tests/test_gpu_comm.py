"""-m gpu: the path's one exchange step (reference units multiclust.c:516-653 initialisations, 681-701 bootstrap
replicates) executed on the one GPU a test box has -- every line of multiclust_amd/csrc/mchip_comm.hip that does not need a
second device: the lazy dlopen of RCCL and its symbols, ncclCommInitAll over one device, the upload / grouped ncclAllReduce /
download sequence on the communicator's own stream for both reductions, buffer growth, the error returns, destruction.
What is left for a multi-GPU node to execute first is RCCL's transport between devices (xGMI), not this code."""
import ctypes as C

import numpy as np
import pytest

from multiclust_amd import hip

pytestmark = pytest.mark.gpu

INVALID, UNSUPPORTED = 1, 6


def make_comm(lib, devices):
    comm = C.c_void_p()
    devs = (C.c_int * len(devices))(*devices)
    return lib.mchip_comm_create(C.byref(comm), len(devices), devs), comm


def all_reduce(lib, comm, tables, op):
    bufs = (C.POINTER(C.c_double) * len(tables))(*[t.ctypes.data_as(C.POINTER(C.c_double)) for t in tables])
    return lib.mchip_comm_all_reduce(comm, bufs, tables[0].size, op)


def test_single_device_communicator_reduces_the_result_table_through_rccl():
    lib = hip.load()
    rc, comm = make_comm(lib, [0])
    assert rc == 0 and comm.value
    n, ver, done = C.c_int(), C.c_int(), C.c_ulonglong(99)
    assert lib.mchip_comm_info(comm, C.byref(n), C.byref(ver), C.byref(done)) == 0
    assert n.value == 1 and done.value == 0
    assert ver.value >= 20000, ver.value                  # ncclGetVersion of the RCCL that was dlopen'ed (2.x.y -> 2xxyy)

    rng = np.random.default_rng(7)
    # the per-unit table of maximize_likelihood (9 fields per initialisation), as mc_main.c fills it
    tab = rng.standard_normal(50 * 9)
    tab[::9] = -3.1e6 - rng.random(50)                   # log likelihoods of config-3 magnitude
    want = tab.copy()
    for op in (0, 1):                                    # sum, max: over ONE device both return the table itself, bit for bit
        got = tab.copy()
        assert all_reduce(lib, comm, [got], op) == 0, lib.mchip_comm_last_error(comm)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    # a longer table re-allocates the device buffer; a shorter one afterwards re-uses it
    big = rng.standard_normal(200 * 2 + 12345)
    keep = big.copy()
    assert all_reduce(lib, comm, [big], 0) == 0
    assert np.array_equal(big, keep)
    small = np.array([1.5, -2.5, float("inf"), 0.0])
    assert all_reduce(lib, comm, [small], 1) == 0
    assert small.tolist() == [1.5, -2.5, float("inf"), 0.0]
    assert lib.mchip_comm_info(comm, None, None, C.byref(done)) == 0 and done.value == 4

    # argument errors leave the communicator usable
    assert all_reduce(lib, comm, [small], 2) == INVALID                       # unknown reduction
    bufs = (C.POINTER(C.c_double) * 1)(small.ctypes.data_as(C.POINTER(C.c_double)))
    assert lib.mchip_comm_all_reduce(comm, bufs, 0, 0) == INVALID             # empty table
    assert lib.mchip_comm_all_reduce(comm, None, 4, 0) == INVALID
    assert lib.mchip_comm_all_reduce(None, bufs, 4, 0) == INVALID
    assert all_reduce(lib, comm, [small], 0) == 0
    assert lib.mchip_comm_destroy(comm) == 0


def test_communicator_argument_errors():
    lib = hip.load()
    have = C.c_int()
    assert lib.mchip_device_count(C.byref(have)) == 0 and have.value >= 1
    rc, comm = make_comm(lib, [have.value])                                   # device index out of range
    assert rc == INVALID and not comm.value
    rc, comm = make_comm(lib, [-1])
    assert rc == INVALID and not comm.value
    comm = C.c_void_p()
    assert lib.mchip_comm_create(C.byref(comm), 0, (C.c_int * 1)(0)) == INVALID
    assert lib.mchip_comm_create(C.byref(comm), 1, None) == INVALID
    assert lib.mchip_comm_create(None, 1, (C.c_int * 1)(0)) == INVALID
    assert lib.mchip_comm_destroy(None) == 0                                  # destroying nothing is not an error
    assert lib.mchip_comm_info(None, None, None, None) == INVALID
    assert b"null" in lib.mchip_comm_last_error(None)


def test_two_communicators_in_one_process():
    """a bootstrap run creates its communicator once per run; nothing in RCCL's state may stop a second run of the same process
    (the Python bindings, a long-lived host) from creating another"""
    lib = hip.load()
    for _ in range(2):
        rc, comm = make_comm(lib, [0])
        assert rc == 0
        t = np.arange(18.0)
        assert all_reduce(lib, comm, [t], 0) == 0 and np.array_equal(t, np.arange(18.0))
        assert lib.mchip_comm_destroy(comm) == 0


COMM_SCRIPT = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
if %d:
    import torch                                     # PyTorch's own librccl.so and HIP runtime are in the process first
    torch.zeros(4, device="cuda").sum().item()
import numpy as np
from multiclust_amd import hip
lib = hip.load()
comm = C.c_void_p()
rc = lib.mchip_comm_create(C.byref(comm), 1, (C.c_int * 1)(0))
assert rc == 0, rc
t = np.arange(27.0)
bufs = (C.POINTER(C.c_double) * 1)(t.ctypes.data_as(C.POINTER(C.c_double)))
assert lib.mchip_comm_all_reduce(comm, bufs, 27, 0) == 0 and np.array_equal(t, np.arange(27.0))
ver = C.c_int()
lib.mchip_comm_info(comm, None, C.byref(ver), None)
rccl = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l]
print("rccl", ver.value, sorted(set(rccl)))
assert len(set(rccl)) == 1, rccl                     # one RCCL in the process, whoever brought it
assert lib.mchip_comm_destroy(comm) == 0
"""


@pytest.mark.parametrize("torch_first", [0, 1])
def test_one_rccl_per_process(torch_first):
    """The communicator in a fresh process, without PyTorch (the command line's situation: ROCm's librccl.so.1 is loaded) and
    with PyTorch imported and its GPU context up first (the Python bindings' situation: PyTorch's librccl.so is already there
    and must be the one used -- a second RCCL beside it made ncclCommInitAll fail)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", COMM_SCRIPT % (root, torch_first)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.split("\n") if l.startswith("rccl ")][-1]
    assert ("torch/lib/librccl.so" in line) == bool(torch_first), line
