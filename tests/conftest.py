import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# what oracle/Makefile builds from /root/reference in the build container and what travels to the GPU box with the tree (the
# directory is git-ignored, not gpurun-ignored): the unmodified reference program, its em() on flat arrays, the harness behind
# the goldens, and the reference's own driver bound to this EM layer.  About 100 of the -m gpu tests compare with them live.
REFERENCE_BINARIES = ("multiclust_ref", "ref_time", "ref_harness", "multiclust_ref_hip")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def missing_reference_binaries(root=ROOT):
    return [b for b in REFERENCE_BINARIES if not os.access(os.path.join(root, "oracle", "_ref", b), os.X_OK)]


def gpu_present():
    if os.environ.get("MC_TEST_ASSUME_GPU"):                 # tests/test_refcheck_cpu.py: the check itself, without a GPU
        return os.environ["MC_TEST_ASSUME_GPU"] == "1"
    try:
        import ctypes as C
        from multiclust_amd import hip
        n = C.c_int(0)
        hip.load().mchip_device_count(C.byref(n))
        return n.value > 0
    except Exception:
        return False


def reference_check(markexpr, gpu, root=ROOT, env=os.environ):
    """None, or why a GPU run must not start: the reference-backed tests would be skipped one by one (`skipif` on the binary) and
    the run would come out green with ninety tests fewer -- a difference only the pass count shows.  On a box with a GPU the
    binaries have to be there unless the caller says otherwise."""
    if "not gpu" in (markexpr or "") or not gpu or env.get("MC_ALLOW_NO_REF") == "1":
        return None
    missing = missing_reference_binaries(root)
    if not missing:
        return None
    return ("oracle/_ref/ lacks %s: the GPU tests that compare with the reference itself would be skipped silently.  Build them where "
            "/root/reference exists (`make -C oracle`; they travel with the tree: oracle/_ref/ must not be listed in .gpurunignore), "
            "or set MC_ALLOW_NO_REF=1 to run without them." % ", ".join(missing))


def pytest_sessionstart(session):
    why = reference_check(session.config.getoption("markexpr", ""), gpu_present(), os.environ.get("MC_TEST_ROOT", ROOT))
    if why:
        raise pytest.UsageError(why)
