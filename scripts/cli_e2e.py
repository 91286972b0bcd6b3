"""End-to-end run of the drop-in command line on a config-2-sized STRUCTURE file (2000 x 20000 biallelic, K = 5):
writes the 160 MB text file, then times `multiclust -a -k 5 -n 1 -r 11 -T <iters>` (read + init + EM + writers)."""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from synth import make_dataset

out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/c2"
iters = sys.argv[2] if len(sys.argv) > 2 else "100"
os.makedirs(out, exist_ok=True)
I, L, K = 2000, 20000, 5
t0 = time.time()
ua, geno = make_dataset(I, L, K, ploidy=2, max_alleles=2, seed=20250119, chunk=256)
path = os.path.join(out, "c2.stru")
with open(path, "w") as f:
    f.write(" ".join("loc%d" % (l + 1) for l in range(L)) + "\n")
    codes = (geno + 1).astype(np.int8)
    for i in range(I):
        for a in range(2):
            f.write("ind%d pop%d " % (i, i % K) + " ".join(map(str, codes[i, :, a])) + "\n")
print("wrote %s (%.0f MB) in %.1f s" % (path, os.path.getsize(path) / 1e6, time.time() - t0), flush=True)
binp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multiclust_amd", "bin", "multiclust")
for extra in (["-T", "1"], ["-T", iters]):
    t0 = time.time()
    r = subprocess.run([binp, "-f", path, "-a", "-k", str(K), "-n", "1", "-r", "11", "-d", out] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    dt = time.time() - t0
    print("args %s: rc=%d wall %.2f s" % (extra, r.returncode, dt))
    print(r.stdout.strip()[:600])
    if r.returncode:
        print(r.stderr[-500:])
