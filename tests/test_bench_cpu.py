"""CPU checks of bench.py's launcher: --gpus N without RANK must start N ranks as a child (never run one rank and print
n_gpus: 1), and must refuse when fewer than N devices are visible."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_needs_that_many_devices():
    """`--gpus 99` (no node has 99 devices): exit code 3, nothing on stdout, and no rank was started -- the check counts
    devices with torch.cuda.device_count(), which does not initialise the GPU on this image, before the child is launched."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MC_BENCH_DEVICE")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "99"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode == 3 and "--gpus 99 but only" in res.stderr and res.stdout == ""
    assert "torch.distributed" not in res.stderr and "Traceback" not in res.stderr


def test_world_size_must_match_gpus_flag():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE" in res.stderr and res.stdout == ""
