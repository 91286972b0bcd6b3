"""A GPU run without the reference binaries must not come out green with ninety tests fewer: tests/conftest.py refuses to start
it (unless MC_ALLOW_NO_REF=1).  Checked here without a GPU: the rule itself, and a pytest session pointed at a tree without them."""
import os
import shutil
import subprocess
import sys

import conftest

ROOT = conftest.ROOT


def fake_tree(tmp_path, binaries):
    d = tmp_path / "oracle" / "_ref"
    d.mkdir(parents=True)
    for b in binaries:
        (d / b).write_text("#!/bin/sh\n")
        os.chmod(str(d / b), 0o755)
    return str(tmp_path)


def test_rule(tmp_path):
    empty = fake_tree(tmp_path / "a", [])
    full = fake_tree(tmp_path / "b", conftest.REFERENCE_BINARIES)
    partial = fake_tree(tmp_path / "c", conftest.REFERENCE_BINARIES[:2])
    assert "multiclust_ref, ref_time, ref_harness, multiclust_ref_hip" in conftest.reference_check("gpu", True, empty, {})
    assert "lacks ref_harness, multiclust_ref_hip" in conftest.reference_check("gpu", True, partial, {})
    assert conftest.reference_check("", True, empty, {}) is not None           # no -m at all also runs the GPU tests
    assert conftest.reference_check("gpu", True, full, {}) is None
    assert conftest.reference_check("gpu", True, empty, {"MC_ALLOW_NO_REF": "1"}) is None
    assert conftest.reference_check("not gpu", True, empty, {}) is None         # the CPU suite needs none of them to start
    assert conftest.reference_check("gpu", False, empty, {}) is None            # no GPU: collection only (this container)
    assert "oracle/_ref/" not in open(os.path.join(ROOT, ".gpurunignore")).read().split()       # ... and they do travel


def test_a_gpu_session_without_the_binaries_does_not_start(tmp_path):
    env = dict(os.environ, MC_TEST_ASSUME_GPU="1", MC_TEST_ROOT=fake_tree(tmp_path, ["multiclust_ref"]))
    env.pop("MC_ALLOW_NO_REF", None)
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_abi.py"), "-q", "-m", "gpu", "-p", "no:cacheprovider"]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 4, (res.returncode, res.stdout[-500:], res.stderr[-500:])          # pytest's usage-error status
    assert "oracle/_ref/ lacks ref_time, ref_harness, multiclust_ref_hip" in res.stderr
    res = subprocess.run(cmd, env=dict(env, MC_ALLOW_NO_REF="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 5, (res.returncode, res.stderr[-500:])                                # started; test_abi.py has no gpu tests
