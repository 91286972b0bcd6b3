"""oracle/_ref/ref_time (the reference's own em() on flat arrays: bench.py's `cpu_baseline` kind "reference") against the CPU
oracle in reference operation order: the same iterate, bit for bit.  The reader bypass of oracle/ref_time.c therefore hands the
reference the data it would have read.  Skipped where the binary is absent (it is built where /root/reference exists)."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_bind as ob
from golden_util import Golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TIME = os.path.join(ROOT, "oracle", "_ref", "ref_time")
pytestmark = pytest.mark.skipif(not os.access(REF_TIME, os.X_OK), reason="oracle/_ref/ref_time not built")


@pytest.mark.parametrize("name", ["c1_admix_k3", "multi_admix_k4", "tetra_admix_k3", "hexaploid_admix_k2",
                                  "multi_admix_c_k3", "multi_mix_k3", "hexaploid_mix_k2"])        # -c and the mixture model: one row of eta
@pytest.mark.parametrize("accel,max_iter", [(0, 5), (3, 7), (1, 9)])
def test_reference_em_on_flat_arrays_equals_the_oracle(tmp_path, name, accel, max_iter):
    g = Golden(name)
    if g.m["eta_constrained"] or not g.m["admixture"]:
        # one row of eta: the fit is at its fixed point to the last bit within a few iterations, and with "never converged"
        # (abs_error 1e-300) the next accelerated cycle ends in the reference's exit on a log likelihood lower by one ulp
        max_iter = min(max_iter, 3)
    I, L, p = g.geno.shape
    d = str(tmp_path)
    g.ua.astype(np.int32).tofile(d + "/ua.i32")
    g.geno.tofile(d + "/geno.u8")
    q0, p0 = g.q("q0"), g.p("p0")
    q0.tofile(d + "/q0.f64")
    p0.tofile(d + "/p0.f64")
    model = (["-a"] if g.m["admixture"] else []) + (["-c"] if g.m["eta_constrained"] else [])
    cmd = [REF_TIME, d, str(I), str(L), str(p), str(g.K), str(max_iter), "--", "-f", "x"] + model + ["-k", str(g.K)]
    res = subprocess.run(cmd + (["-s", str(accel)] if accel else []), capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    j = json.loads(res.stdout)
    opt = ob.make_options(admixture=g.m["admixture"], eta_constrained=g.m["eta_constrained"], lower_bound=j["lower_bound"], fused=0,
                          accel_scheme=accel, abs_error=1e-300, max_iter=max_iter)
    mod = ob.Model(ob.Data(I, L, p, g.ua, g.geno), opt, g.K)
    mod.q(0)[...] = q0
    mod.p(0)[...] = p0
    mod.em()
    assert j["n_iter"] == mod.n_iter <= max_iter + 1          # (-c and the mixture model reach their fixed point to the last bit earlier)
    assert j["iter_stop"] == int(mod.n_iter == max_iter + 1)
    assert j["logL"] == mod.logL
    assert np.array_equal(np.fromfile(d + "/q_ref.f64").reshape(mod.q(mod.pindex).shape), mod.q(mod.pindex))
    assert np.array_equal(np.fromfile(d + "/p_ref.f64").reshape(g.K, -1), mod.p(mod.pindex))


@pytest.mark.parametrize("name", ["missing_admix_k3", "allmiss_admix_k2", "triploid_admix_k3", "missing_mix_k2"])
def test_missing_copies_count_nowhere(tmp_path, name):
    """data with missing values, phantom allele slots included in ua[] as the reference's reader counts them: the same iterate as
    the oracle, bit for bit (allmiss: loci at which every copy is missing have no allele column at all -- ref_time refuses a
    locus without alleles, so that fixture is expected to be refused)"""
    g = Golden(name)
    I, L, p = g.geno.shape
    d = str(tmp_path)
    g.ua.astype(np.int32).tofile(d + "/ua.i32")
    g.geno.tofile(d + "/geno.u8")
    q0, p0 = g.q("q0"), g.p("p0")
    q0.tofile(d + "/q0.f64")
    p0.tofile(d + "/p0.f64")
    model = (["-a"] if g.m["admixture"] else []) + (["-c"] if g.m["eta_constrained"] else [])
    res = subprocess.run([REF_TIME, d, str(I), str(L), str(p), str(g.K), "3", "--", "-f", "x"] + model + ["-k", str(g.K)],
                         capture_output=True, text=True, timeout=120)
    if name.startswith("allmiss"):
        assert res.returncode == 2 and "locus without alleles" in res.stderr
        return
    assert res.returncode == 0, res.stderr
    j = json.loads(res.stdout)
    opt = ob.make_options(admixture=g.m["admixture"], eta_constrained=g.m["eta_constrained"], lower_bound=j["lower_bound"], fused=0,
                          abs_error=1e-300, max_iter=3)
    mod = ob.Model(ob.Data(I, L, p, g.ua, g.geno), opt, g.K)
    mod.q(0)[...] = q0
    mod.p(0)[...] = p0
    mod.em()
    assert j["n_iter"] == mod.n_iter and j["logL"] == mod.logL
    assert np.array_equal(np.fromfile(d + "/q_ref.f64").reshape(mod.q(mod.pindex).shape), mod.q(mod.pindex))
    assert np.array_equal(np.fromfile(d + "/p_ref.f64").reshape(g.K, -1), mod.p(mod.pindex))
