"""one-off: the drawn case on which the command line did not return (mixture, haploid, K = 2, -s 3 -g 3): rebuild its data file and
run both programs with -v 4 under a short time limit"""
import os, random, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from make_fixtures import write_stru
I, L, ploidy, K, seed = 31, 60, 1, 2, 4822960
seed, missing = seed // 10, (seed % 10) / 100.0
rnd = random.Random(seed)
alleles = [2, 2, 3, 4, 5, 12, 36] if seed % 5 == 0 else [2, 2, 3, 4, 5]
os.makedirs("/tmp/h184", exist_ok=True)
stru = "/tmp/h184/d.stru"
write_stru(stru, I, L, max(2, K - 1), ploidy, [rnd.choice(alleles) for _ in range(L)], seed=seed, missing=missing)
args = ["-p", "1", "-k", "2", "-r", str(seed % 9973 + 1), "-n", "2", "-g", "3", "-T", "40", "-s", "3", "-v", "4"]
for name, exe in (("ref", os.path.join(ROOT, "oracle/_ref/multiclust_ref")), ("hip", os.path.join(ROOT, "multiclust_amd/bin/multiclust"))):
    d = "/tmp/h184/" + name
    os.makedirs(d, exist_ok=True)
    try:
        res = subprocess.run([exe, "-f", stru, "-d", d + "/"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=15, cwd=d)
        print("==", name, "rc", res.returncode); print(res.stdout[-600:]); print(res.stderr[-1500:])
    except subprocess.TimeoutExpired as e:
        print("==", name, "TIMEOUT"); print((e.stdout or b"")[-600:]); print((e.stderr or b"")[-2500:].decode() if isinstance(e.stderr, bytes) else str(e.stderr)[-2500:])
