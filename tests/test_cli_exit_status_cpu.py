"""The command line's exit status on the failures that happen before a GPU is needed -- bad command lines, unreadable or
malformed data files, impossible set-ups -- against the unmodified reference program (oracle/_ref/multiclust_ref, where it has
been built): the reference returns its error enum from main() (message.h:21-41), so `multiclust ... || handle $?` keeps working
after the switch.  Runs without a GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "multiclust_amd", "bin", "multiclust")
REFBIN = os.path.join(ROOT, "oracle", "_ref", "multiclust_ref")
MULTI = os.path.join(ROOT, "tests", "golden", "data", "multi.stru")

CASES = [
    (["-a", "-k", "2"], 8),                                  # no -f: INVALID_CMDLINE
    (["-f", "/nonexistent/file.stru", "-a", "-k", "2"], 5),  # FILE_OPEN_ERROR
    (["-f", MULTI, "-a", "-k", "200"], 11),                  # more clusters than individuals: INVALID_USER_SETUP
    (["-f", MULTI, "-a", "-1", "3", "-2", "2"], 11),         # minimum K above maximum K
    (["-f", MULTI, "-a", "-k", "1", "-b", "2"], 11),         # bootstrap needs K > 1
    (["-f", MULTI, "-a", "-k", "2", "-Z"], 9),               # INVALID_CMD_OPTION
    (["-f", MULTI, "-a", "-k", "x"], 10),                    # INVALID_CMD_ARGUMENT
    (["-f", MULTI, "-a", "-k", "2", "-T", "-3"], 10),
    (["-f", MULTI, "-p", "3", "-a", "-k", "2"], 7),          # 80 lines are not a multiple of ploidy 3: FILE_FORMAT_ERROR
    (["-h"], 1),                                             # help leaves through the usage-error path
    (["-f", MULTI, "-a", "-k", "2", "-A", "/nonexistent/partition.txt"], 5),     # -A: read_afile() is synchronize()'s last step
    (["-f", MULTI, "-a", "-k", "2", "-A", MULTI], 7),        # ... a file that does not start with I integers: FILE_FORMAT_ERROR
]


@pytest.mark.parametrize("args,status", CASES)
def test_exit_status(args, status, tmp_path):
    ours = subprocess.run([BIN] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60, cwd=str(tmp_path))
    assert ours.returncode == status, (args, ours.returncode, ours.stderr[-500:])
    if os.access(REFBIN, os.X_OK):
        ref = subprocess.run([REFBIN] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60, cwd=str(tmp_path))
        assert ref.returncode == status, (args, ref.returncode)
