"""Moderate-size exercise of the command line paths that the small golden fixtures only touch lightly: several
initialisations with SQUAREM, a K range, a bootstrap (device-generated replicates), and the sharded rehearsal."""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from synth import make_dataset

out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/stress"
os.makedirs(out, exist_ok=True)
I, L, K = 600, 6000, 3
ua, geno = make_dataset(I, L, K, ploidy=2, max_alleles=4, seed=5, missing=0.01)
path = os.path.join(out, "s.stru")
with open(path, "w") as f:
    f.write(" ".join("loc%d" % (l + 1) for l in range(L)) + "\n")
    codes = np.where(geno == 255, -9, geno.astype(np.int16) + 101)
    for i in range(I):
        for a in range(2):
            f.write("ind%d pop%d " % (i, i % K) + " ".join(map(str, codes[i, :, a])) + "\n")
binp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multiclust_amd", "bin", "multiclust")
runs = [
    (["-a", "-k", "3", "-n", "4", "-s", "3", "-r", "3"], {}),
    (["-a", "-1", "2", "-2", "4", "-n", "2", "-s", "3", "-r", "3"], {}),
    (["-a", "-k", "3", "-n", "2", "-b", "3", "-s", "3", "-r", "3"], {}),
    (["-a", "-k", "3", "-n", "2", "-b", "3", "-s", "3", "-r", "3", "--gpus", "1"], {"MC_FORCE_SHARDED": "1"}),
    (["-a", "-k", "3", "-n", "2", "-b", "3", "-s", "3", "-r", "3"], {"MC_HOST_BOOTSTRAP": "1"}),
    (["-k", "3", "-n", "2", "-b", "2", "-r", "3"], {}),
]
tails = []
for args, env in runs:
    t0 = time.time()
    r = subprocess.run([binp, "-f", path, "-d", out] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, **env))
    lines = r.stdout.strip().split("\n")
    print("args %s env %s: rc=%d wall %.1f s, %d stdout lines; last: %s" % (args, env, r.returncode, time.time() - t0, len(lines), lines[-1][:160]), flush=True)
    if r.returncode:
        print(r.stderr[-800:])
        sys.exit(1)
    tails.append([l for l in lines if "test statistics" in l or "p-value" in l])
assert tails[2] == tails[3] == tails[4], (tails[2], tails[3], tails[4])
# several fits at a time on one GPU: same summary line, less wall time on a data set this small
finals = []
for streams in (1, 4, 8):
    t0 = time.time()
    r = subprocess.run([binp, "-f", path, "-d", out, "-a", "-k", "3", "-n", "32", "-s", "3", "-r", "3", "--streams", str(streams)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-500:]
    print("32 initialisations, --streams %d: wall %.1f s" % (streams, time.time() - t0), flush=True)
    finals.append(r.stdout.strip().split("\n")[-1].split()[:12])
assert finals[0] == finals[1] == finals[2], finals
print("bootstrap: device, sharded and host-drawn runs print the same test statistics")
