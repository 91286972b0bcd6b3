"""multiclust_amd -- MI355X-native EM hot path of MULTICLUST behind a C-ABI.

The product is two shared libraries built in-tree by `make` (see __graft_entry__.build()):
  lib/libmulticlust_hip.so   hand-written HIP kernels for gfx950 + the C-ABI (include/multiclust_hip.h)
  lib/libmulticlust_host.so  plain-C host side mirroring the reference's em()/stop()/accelerated_em_step()
This package is only the thin ctypes view of those libraries used by tests/ and bench.py.
There is no CPU fallback: loading fails loudly when the HIP library is missing.
"""
from .hip import Context, HipError, lib_path, load  # noqa: F401
