"""The CPU oracle (oracle/mc_oracle.c in reference operation order) against the reference's own em() (oracle/_ref/ref_time), live,
on drawn shapes: ploidy 1-6, up to 12 alleles per locus, K 1-12, admixture / -c / mixture, plain EM and every acceleration
scheme.  Same iterate BIT FOR BIT -- log likelihood, mixing proportions, allele frequencies, iteration count -- which is what lets
the GPU tests use the oracle as the reference's stand-in on shapes no committed golden covers (tests/test_gpu_fuzz.py).
A quarter of the cases carry 3 % missing copies, a quarter 25 %, most of them with the extra allele slot the reference's reader
gives a locus with missing values.
Runs without a GPU; skipped where the binary is absent."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_bind as ob
from synth import make_dataset, random_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TIME = os.path.join(ROOT, "oracle", "_ref", "ref_time")
pytestmark = pytest.mark.skipif(not os.access(REF_TIME, os.X_OK), reason="oracle/_ref/ref_time not built")


def draw_cases(n, seed):
    rs = np.random.default_rng(seed)
    out = []
    for c in range(n):
        out.append((c, int(rs.integers(3, 70)), int(rs.integers(1, 90)), int(rs.choice([1, 2, 2, 2, 3, 4, 6])), int(rs.choice([2, 3, 4, 6, 12])),
                    int(rs.choice([1, 2, 3, 4, 5, 8, 12])), str(rs.choice(["admix", "admix", "admix_c", "mix"])),
                    int(rs.choice([0, 0, 1, 2, 3, 4, 5, 6])), int(rs.integers(2, 12)), int(rs.integers(1, 10 ** 6))))
    return out


@pytest.mark.parametrize("c,I,L,ploidy,maxal,K,model,scheme,iters,seed",
                         draw_cases(int(os.environ.get("MC_ORACLE_CASES", "60")), 99 + int(os.environ.get("MC_ORACLE_SEED", "0"))))
def test_oracle_against_the_reference_em_bit_for_bit(c, I, L, ploidy, maxal, K, model, scheme, iters, seed, tmp_path):
    I = max(I, K)
    missing = [0.0, 0.0, 0.03, 0.25][seed % 4]
    ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=maxal, seed=seed, missing=missing)
    if missing and seed % 8 < 6:
        # as the reference's reader shapes such data: a locus with missing copies counts one allele slot more, which nothing matches
        ua = (ua + (geno == 0xFF).any(axis=(0, 2)).astype(np.int32)).astype(np.int32)
    admixture, constrained = int(model != "mix"), int(model == "admix_c")
    if K == 1:
        scheme = 0                                         # em() leaves before any acceleration (em_alg.c:49-58)
    lb = min(1e-8, 0.5 / (I * ploidy))
    q0, p0 = random_params(I, ua, K, seed=seed + 1, lower_bound=lb)
    if constrained or not admixture:
        q0 = np.ascontiguousarray(q0.mean(axis=0) / q0.mean(axis=0).sum())
    d = str(tmp_path)
    np.ascontiguousarray(ua, dtype=np.int32).tofile(d + "/ua.i32")
    geno.tofile(d + "/geno.u8")
    q0.tofile(d + "/q0.f64")
    p0.tofile(d + "/p0.f64")
    flags = (["-a"] if admixture else []) + (["-c"] if constrained else []) + ["-k", str(K)] + (["-s", str(scheme)] if scheme else [])
    res = subprocess.run([REF_TIME, d, str(I), str(L), str(ploidy), str(K), str(iters - 1), "--", "-f", "x"] + flags,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    opt = ob.make_options(admixture=admixture, eta_constrained=constrained, lower_bound=lb, fused=0, accel_scheme=scheme,
                          abs_error=1e-300, max_iter=iters - 1)
    mod = ob.Model(ob.Data(I, L, ploidy, ua, geno), opt, K)
    mod.q(0)[...] = q0
    mod.p(0)[...] = p0
    mod.em()
    if res.returncode == 0 and not res.stdout.strip():
        # the reference left through exit(0): "nan", or a log likelihood one ulp lower at a fixed point reached to the last bit;
        # the oracle reports the same ending as a fatal state
        assert mod.fatal != 0, res.stderr[-300:]
        return
    assert res.returncode == 0, res.stderr[-1000:]
    ref = json.loads(res.stdout)
    assert mod.fatal == 0 and ref["n_iter"] == mod.n_iter
    assert ref["lower_bound"] == lb and ref["logL"] == mod.logL
    # (equal_nan: an individual whose every copy is missing -- a one-locus draw at 25 % -- has mixing proportions 0 / 0 in both)
    assert np.array_equal(np.fromfile(d + "/q_ref.f64").reshape(mod.q(mod.pindex).shape), mod.q(mod.pindex), equal_nan=True)
    assert np.array_equal(np.fromfile(d + "/p_ref.f64").reshape(K, -1), mod.p(mod.pindex), equal_nan=True)
