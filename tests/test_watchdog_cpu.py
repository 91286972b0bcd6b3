"""No GPU needed: the record of where the process stands (mchip_progress_note / mchip_progress_report, include/multiclust_hip.h)
and the opt-in watchdog of the command line (MC_WATCHDOG_S, multiclust_amd/host/mc_watchdog.c): a process that makes no progress
says where it stands and leaves with status 3 -- it is never launched again (tests/procutil.py)."""
import ctypes as C
import os
import subprocess
import sys
import time

import pytest

import procutil
from multiclust_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTLIB = os.path.join(ROOT, "multiclust_amd", "lib", "libmulticlust_host.so")

STALL = r"""
import ctypes as C, sys, time
sys.path.insert(0, %r)
from multiclust_amd import hip
lib = hip.load()
host = C.CDLL(%r)
host.mc_watchdog_start.argtypes = [C.c_double]
print("first line", flush=True)
lib.mchip_progress_note(b"phase one")
for _ in range(5):                      # a process that keeps reporting events is left alone
    time.sleep(0.3)
    lib.mchip_progress_note(b"phase one")
host.mc_watchdog_start(0.6)
for _ in range(4):
    time.sleep(0.3)
    lib.mchip_progress_note(b"still moving")
print("second line", flush=True)
lib.mchip_progress_note(b"the phase that never ends")
time.sleep(30)                          # ... and one that stops is ended
print("not reached", flush=True)
"""


def test_progress_record_names_the_phase_and_counts_events():
    lib = hip.load()
    ev0, ev1 = C.c_ulonglong(), C.c_ulonglong()
    assert lib.mchip_progress_report(None, 0, C.byref(ev0)) == 0
    note = b"test_watchdog_cpu: a phase"
    assert lib.mchip_progress_note(note) == 0
    buf = C.create_string_buffer(2048)
    assert lib.mchip_progress_report(buf, 2048, C.byref(ev1)) == 0
    assert ev1.value == ev0.value + 1
    text = buf.value.decode()
    assert "outside the library" in text and "last phase: test_watchdog_cpu: a phase" in text and "thread %d" % os.getpid() in text
    small = C.create_string_buffer(16)                      # any length: truncated, terminated
    assert lib.mchip_progress_report(small, 16, None) == 0 and len(small.value) <= 15
    assert lib.mchip_device_count(C.byref(C.c_int())) == 0  # an entry point counts two events (in, out) plus its runtime call
    lib.mchip_progress_report(None, 0, C.byref(ev0))
    assert ev0.value >= ev1.value + 4


def test_watchdog_ends_a_process_that_stands_still_and_says_where():
    t0 = time.time()
    res = subprocess.run([sys.executable, "-c", STALL % (ROOT, HOSTLIB)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert res.returncode == 3, (res.returncode, res.stderr)
    assert time.time() - t0 < 20
    assert res.stdout == "first line\nsecond line\n"
    assert "WATCHDOG [mc_watchdog.c]: no progress for" in res.stderr
    assert "last phase: the phase that never ends" in res.stderr
    assert "task " in res.stderr and "state S" in res.stderr        # the kernel's view of the threads: the main one sleeps


def test_launcher_fails_a_program_at_its_limit_with_the_evidence_and_launches_once(tmp_path):
    marker = tmp_path / "launches"
    script = "import time; open(%r, 'a').write('x'); print('so far', flush=True); time.sleep(60)" % str(marker)
    with pytest.raises(procutil.ProgramTimeout) as e:
        procutil.run_program([sys.executable, "-c", script], cwd=str(tmp_path), timeout=2)
    msg = str(e.value)
    assert "no return within 2 s" in msg and "ended on SIGTERM" in msg
    assert "stdout so far:\nso far" in msg and "state S" in msg and "wait channel" in msg
    assert marker.read_text() == "x"                                  # one launch
    ok = procutil.run_program([sys.executable, "-c", "import os; print(os.environ['MC_WATCHDOG_S'])"], timeout=120)
    assert ok.returncode == 0 and ok.stdout.strip() == "60"           # ours run under the watchdog, below the limit
