"""Bit-level fingerprint of the EM path on a few seeded data sets (diploid / tetraploid, with and without missing data, ragged
sizes): run it under two builds of the library (MCHIP_LIB_PATH) and diff the output.  Used to check that a re-scheduled kernel
computes the same bits as the shipped one (scripts/diag/bits.sh)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiclust_amd as mc
from multiclust_amd import host
from synth import make_dataset, random_params

Ks = [int(k) for k in os.environ.get("BITS_K", "8").split(",")]
for K in Ks:
    for (I, L, ploidy, maxal, missing) in [(300, 1003, 2, 4, 0.0), (257, 515, 2, 3, 0.03), (1000, 4096, 2, 4, 0.0), (130, 77, 2, 2, 0.0),
                                          (200, 301, 4, 4, 0.0), (64, 8, 2, 4, 0.1)]:
        ua, geno = make_dataset(I, L, K, ploidy=ploidy, max_alleles=maxal, seed=11 + I, missing=missing)
        for accel in (0, 3):
            fit = host.Fit(ua, geno, K, admixture=1, accel_scheme=accel, verbosity=1, abs_error=1e-300)
            q0, p0 = random_params(I, ua, K, seed=5, lower_bound=fit.opt.lower_bound)
            fit.set_params(q0, p0)
            fit.opt.max_iter = 11
            fit.em()                                     # batched runs (dual pass where it applies)
            h = hashlib.sha256()
            h.update(np.float64(fit.mod.logL).tobytes())
            h.update(fit.get_q(fit.mod.pindex).tobytes())
            h.update(fit.get_p(fit.mod.pindex).tobytes())
            h.update(fit.expected_counts().tobytes())
            print("K=%d I=%d L=%d p=%d M<=%d miss=%g s=%d: n_iter=%d logL=%s %s" % (
                K, I, L, ploidy, maxal, missing, accel, fit.mod.n_iter, float(fit.mod.logL).hex(), h.hexdigest()[:16]), flush=True)
            fit.close()
