"""-m gpu: Rand-EM initialisation (rnd_init.c:123-160, 412-444, 496-705) -- the host walks the loci and draws the center
alleles, the device assigns every allele copy, counts, normalises and scores each candidate with one EM iteration -- against
the reference's own routine (harness dumps with initialization_procedure = RAND_EM) and, on larger seeded inputs, the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from procutil import run_program

import oracle_bind as ob
from golden_util import GOLD, RANDEM_CASES, Golden
from multiclust_amd import host
from synth import make_dataset

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "multiclust_amd", "bin", "multiclust")


def randem_fit(g, n, **kw):
    return host.Fit(g.ua, g.geno, g.K, admixture=g.m["admixture"], eta_constrained=g.m["eta_constrained"], verbosity=1,
                    initialization_procedure=1, n_rand_em_init=n, **kw)


@pytest.mark.parametrize("name", RANDEM_CASES)
def test_randem_initialisation_vs_reference(name):
    g = Golden(name)
    fit = randem_fit(g, g.m["randem_candidates"])
    rng = fit.initialize(g.m["seed"])
    assert fit.lib.mc_rand(rng) == g.m["rand_after_randem"]            # the stream stands where the reference's rand() does
    # a candidate's parameters are ratios of small integers: identical; which candidate wins is decided by log likelihoods
    # that differ by whole units
    np.testing.assert_allclose(fit.get_q(0), g.q("randem"), rtol=1e-15, atol=0)
    np.testing.assert_allclose(fit.get_p(0), g.p("randem"), rtol=1e-15, atol=0)
    fit.em()
    m = fit.mod
    assert m.fatal == 0 and m.converged == g.m["randem_run_converged"]
    assert abs(m.n_iter - g.m["randem_run_n_iter"]) <= 1
    assert abs(m.logL - g.m["randem_run_logL"]) <= 2e-4
    if m.n_iter == g.m["randem_run_n_iter"]:
        assert abs(m.logL - g.m["randem_run_logL"]) <= 1e-8
        np.testing.assert_allclose(fit.get_q(m.pindex), g.q("randemrun"), rtol=1e-6, atol=1e-10)
        np.testing.assert_allclose(fit.get_p(m.pindex), g.p("randemrun"), rtol=1e-6, atol=1e-10)
    fit.close()


@pytest.mark.parametrize("name", [n for n in RANDEM_CASES if Golden(n).m["admixture"]])
def test_randem_single_candidate_is_the_reference_first_candidate(name):
    g = Golden(name)
    fit = randem_fit(g, 1)
    fit.initialize(g.m["seed"])
    np.testing.assert_allclose(fit.get_q(0), g.q("randem_c0"), rtol=1e-15, atol=0)
    np.testing.assert_allclose(fit.get_p(0), g.p("randem_c0"), rtol=1e-15, atol=0)
    fit.close()


@pytest.mark.parametrize("I,L,ploidy,K,maxal,missing,constrained", [
    (300, 700, 2, 4, 12, 0.03, 0),       # most loci have more alleles than clusters: center draws, retries, many unmatched copies
    (257, 513, 4, 3, 6, 0.0, 0),         # tetraploid, no missing data
    (120, 900, 2, 8, 4, 0.02, 0),        # config-3-like: fewer alleles than clusters everywhere, only missing copies draw
    (150, 400, 3, 5, 9, 0.05, 1),        # odd ploidy, shared mixing proportions
    (64, 100, 2, 1, 5, 0.1, 0),          # K = 1: nothing is drawn
])
def test_randem_candidates_vs_oracle(I, L, ploidy, K, maxal, missing, constrained):
    """Three candidates on a seeded data set: same winner, same parameters, same stream position as the oracle's serial
    restatement of random_allele_center / initialize_parameters_admixture / em_e_step."""
    ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=maxal, seed=3 * I + L, missing=missing)
    n = 3
    fit = host.Fit(ua, geno, K, admixture=1, eta_constrained=constrained, verbosity=1, initialization_procedure=1, n_rand_em_init=n)
    rng = fit.initialize(4242)
    opt = ob.make_options(eta_constrained=constrained, lower_bound=fit.opt.lower_bound, fused=1)
    mod = ob.Model(ob.Data(I, L, ploidy, ua, geno), opt, K)
    org, ll = mod.init_randem(4242, n)
    assert fit.lib.mc_rand(rng) == ob.lib.mco_rand(org)
    np.testing.assert_allclose(fit.get_q(0), mod.q(0), rtol=1e-15, atol=0)
    np.testing.assert_allclose(fit.get_p(0), mod.p(0), rtol=1e-15, atol=0)
    fit.close()


@pytest.mark.parametrize("name,admix", [("multi_admix_k3_randem", 1), ("missing_admix_k2_randem", 1), ("multi_mix_k3_randem", 0)])
def test_skipping_initialisations_moves_the_stream_like_doing_them(name, admix):
    """mc_skip_initializations: where unit n of a sharded run starts.  The draws of a Rand-EM initialisation depend on the
    draws themselves and on the data, so the skip replays the host-side walk: same stream position as n real initialisations."""
    g = Golden(name)
    for randem in (1, 0):
        fit = host.Fit(g.ua, g.geno, g.K, admixture=admix, verbosity=1, initialization_procedure=randem, n_rand_em_init=3)
        done, skipped = host.McRng(), host.McRng()
        fit.lib.mc_srand(C.byref(done), 99)
        fit.lib.mc_srand(C.byref(skipped), 99)
        for _ in range(2):
            assert not fit.lib.mc_initialize_model(C.byref(fit.opt), C.byref(fit.dat), fit.mp, C.byref(done))
        assert not fit.lib.mc_skip_initializations(C.byref(fit.opt), C.byref(fit.dat), fit.mp, C.byref(skipped), 2)
        assert [fit.lib.mc_rand(C.byref(done)) for _ in range(4)] == [fit.lib.mc_rand(C.byref(skipped)) for _ in range(4)]
        fit.close()


def test_cli_randem_serial_sharded_and_reference_fit(tmp_path, monkeypatch):
    """--randem -m 6: the first initialisation's fit is the one the reference's own Rand-EM + em() reaches (harness), and the
    sharded path (units start where mc_skip_initializations says) prints what the serial loop prints."""
    g = Golden("multi_admix_k3_randem")
    cmd = [BIN, "-f", os.path.join(GOLD, "data", "multi.stru"), "-d", str(tmp_path), "-a", "-k", "3", "-r", "7", "-n", "3",
           "--randem", "-m", str(g.m["randem_candidates"])]
    outs = []
    for sharded in (0, 1):
        monkeypatch.delenv("MC_FORCE_SHARDED", raising=False)
        extra = []
        if sharded:
            monkeypatch.setenv("MC_FORCE_SHARDED", "1")
            extra = ["--gpus", "1", "--streams", "2"]
        res = run_program(cmd + extra, timeout=300)
        assert res.returncode == 0, res.stderr
        outs.append(re.sub(r"\d\d:\d\d:\d\d", "HH:MM:SS", res.stdout))
    first = re.search(r"initialization = 0: (-?\d+\.\d+) \((\w+ ?\w*)\) in\s+(\d+) iterations", outs[0])
    assert abs(float(first.group(1)) - g.m["randem_run_logL"]) <= 2e-6 and abs(int(first.group(3)) - g.m["randem_run_n_iter"]) <= 1
    assert outs[0] == outs[1]


def test_cli_randem_with_bootstrap_and_mixture(tmp_path, monkeypatch):
    """--randem together with -b (the replicates' fits initialise from the observed haplotypes, candidates included: the
    reference's random_allele_center reads dat->IL) and for the mixture model: the runs complete, and sharding the
    initialisations inside each replicate prints what the serial loop prints."""
    stru = os.path.join(GOLD, "data", "multi.stru")
    outs = []
    for sharded in (0, 1):
        monkeypatch.delenv("MC_FORCE_SHARDED", raising=False)
        extra = []
        if sharded:
            monkeypatch.setenv("MC_FORCE_SHARDED", "1")
            extra = ["--gpus", "1"]
        res = run_program([BIN, "-f", stru, "-d", str(tmp_path), "-a", "-k", "3", "-r", "5", "-n", "2", "-b", "2", "--randem", "-m", "3", "-T", "30"] + extra,
                          timeout=600)
        assert res.returncode == 0, res.stderr
        assert res.stdout.count("Bootstrap dataset") == 2 and "p-value to reject H0: K=2" in res.stdout
        outs.append(re.sub(r"\d\d:\d\d:\d\d", "HH:MM:SS", res.stdout))
    assert outs[0] == outs[1]
    res = run_program([BIN, "-f", stru, "-d", str(tmp_path), "-k", "3", "-r", "5", "-n", "2", "--randem", "-m", "4"], timeout=600)
    assert res.returncode == 0, res.stderr
    g = Golden("multi_mix_k3_randem")           # same file, seed and model; 5 candidates there, 4 here: only the shape of the output
    assert res.stdout.count("initialization =") == 2 and "converged" in res.stdout
