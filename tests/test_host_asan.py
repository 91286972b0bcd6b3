"""The plain-C host side under AddressSanitizer + UBSan (CPU build): reader on every fixture (including the error
paths), writers, partition, rand() jump-ahead, the threaded partition draw and the bookkeeping."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_host_c_code_is_sanitizer_clean(tmp_path):
    exe = str(tmp_path / "asan_host")
    host = os.path.join(ROOT, "multiclust_amd", "host")
    srcs = [os.path.join(ROOT, "tests", "asan_host_driver.c")] + [os.path.join(host, f) for f in
                                                                  ("mc_reader.c", "mc_writer.c", "mc_fit.c", "mc_em.c", "mc_init.c", "mc_watchdog.c")]
    subprocess.run(["gcc", "-std=gnu11", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    "-I" + os.path.join(ROOT, "include"), "-I" + host, "-o", exe] + srcs + ["-lm", "-lpthread"], check=True)
    data = os.path.join(ROOT, "tests", "golden", "data")
    args = []
    for fn, p in (("multi.stru", 2), ("missing.stru", 2), ("tetra.stru", 4), ("multi_interleaved.stru", 2),
                  ("c1_tiny.stru", 2), ("multi.stru", 3), ("missing99.stru", 2)):
        args += [os.path.join(data, fn), str(p)]
    os.makedirs("/tmp/asan", exist_ok=True)
    res = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1"))
    assert res.returncode == 0, res.stderr[-2000:]
    assert "ok 1" in res.stdout and "AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr
    assert "of a partition with itself 1.000000" in res.stdout and "WATCHDOG [mc_watchdog.c]" in res.stdout
