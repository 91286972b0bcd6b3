"""Launching the command-line programs from the tests: one launch, never a second one.

A launch that does not come back within its limit FAILS the test, with the evidence in the failure: which program it was, its
arguments, what it had written to stdout (line-buffered in ours) and stderr, and the kernel's view of every thread of the process
(/proc/<pid>/task/*: name, state, wait channel; /proc/<pid>/stack where readable) taken BEFORE anything is sent to it; then SIGTERM,
and SIGKILL only after five more seconds.  Our own program runs with MC_WATCHDOG_S set below the limit (multiclust_amd/host/
mc_watchdog.c), so a launch that stops making progress normally ends by itself with status 3 and a report of the library call
each of its threads stands in -- which the caller's `assert res.returncode == 0, res.stderr` then shows.

(Round 3 wrapped these launches in a retry after one of about 4 000 did not return within 300 s; the retry is gone: it hid the
next one.  DESIGN.md section 9.)"""
import os
import signal
import subprocess


class ProgramTimeout(AssertionError):
    pass


def _read(path, limit=4000):
    try:
        with open(path) as f:
            return f.read(limit).strip()
    except OSError as e:
        return "<%s>" % e.strerror


def thread_states(pid):
    """name, state and wait channel of every thread of `pid`, one line each (what hang_stress.sh used to record)"""
    out = []
    base = "/proc/%d/task" % pid
    try:
        tids = sorted(os.listdir(base), key=int)
    except OSError as e:
        return "<no /proc/%d/task: %s>" % (pid, e.strerror)
    for tid in tids:
        status = _read(os.path.join(base, tid, "status"))
        state = next((ln.split(":", 1)[1].strip() for ln in status.split("\n") if ln.startswith("State:")), "?")
        out.append("  task %s (%s): state %s, wait channel %s" % (tid, _read(os.path.join(base, tid, "comm")), state,
                                                                  _read(os.path.join(base, tid, "wchan")) or "0"))
    out.append("  /proc/%d/stack: %s" % (pid, _read("/proc/%d/stack" % pid).replace("\n", " | ")))
    return "\n".join(out)


def run_program(cmd, cwd=None, timeout=120, env=None, watchdog=True):
    """subprocess.run(cmd, stdout=PIPE, stderr=PIPE, text=True) with the timeout behaviour described above"""
    env = dict(os.environ if env is None else env)
    if watchdog:
        env.setdefault("MC_WATCHDOG_S", "%d" % max(10, min(100, timeout // 2)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=cwd, env=env)
    try:
        out, err = proc.communicate(timeout=timeout)
        return subprocess.CompletedProcess(cmd, proc.returncode, out, err)
    except subprocess.TimeoutExpired:
        evidence = thread_states(proc.pid)           # before any signal: where it stands, not how it dies
        proc.send_signal(signal.SIGTERM)
        try:
            out, err = proc.communicate(timeout=5)
            ending = "ended on SIGTERM with status %s" % proc.returncode
        except subprocess.TimeoutExpired:
            proc.kill()
            out, err = proc.communicate()
            ending = "ignored SIGTERM for 5 s, killed"
        raise ProgramTimeout("no return within %d s (%s): %s\n  cwd: %s\n  threads at the limit:\n%s\n  stdout so far:\n%s\n  stderr so far:\n%s"
                             % (timeout, ending, " ".join(cmd), cwd, evidence, (out or "")[-3000:], (err or "")[-6000:])) from None
