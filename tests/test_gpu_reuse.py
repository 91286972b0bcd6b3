"""-m gpu: a context that receives a data set of the same shape keeps its buffers (and parks its model's); a data set
copied between contexts on the device; the cell counts.  Every result must equal what a fresh context gives."""
import numpy as np
import pytest

import multiclust_amd as mc
import oracle_bind as ob
from synth import make_dataset, random_params

pytestmark = pytest.mark.gpu


def fresh_result(ua, geno, K, q0, p0, lb, **kw):
    c = mc.Context(0)
    c.set_genotypes(ua, geno)
    c.set_model(K, lower_bound=lb, **kw)
    c.set_q(0, q0)
    c.set_p(0, p0)
    out = [c.em_step(0, 0) for _ in range(3)] + [c.loglik(0)], c.get_q(0), c.get_p(0), c.expected_counts(), c.data_counts()
    c.close()
    return out


def same(a, b):
    return a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:4], b[1:4])) and a[4] == b[4]


def test_same_shape_data_sets_reuse_the_context():
    I, L, K = 200, 600, 5
    ua, g1 = make_dataset(I, L, K, ploidy=2, max_alleles=4, seed=1, missing=0.0)
    _, g2 = make_dataset(I, L, K, ploidy=2, max_alleles=4, seed=2, missing=0.03)      # same allele counts? regenerate with ua below
    rs = np.random.default_rng(5)
    g2 = (rs.integers(0, 1 << 20, size=(I, L, 2)) % ua[None, :, None]).astype(np.uint8)
    g2[rs.random(g2.shape) < 0.03] = 0xFF                                            # missing copies: another kernel variant
    lb = ob.lib.mco_lower_bound(1e-8, I, 2)
    q0, p0 = random_params(I, ua, K, seed=3, lower_bound=lb)
    ref1 = fresh_result(ua, g1, K, q0, p0, lb)
    ref2 = fresh_result(ua, g2, K, q0, p0, lb)
    c = mc.Context(0)
    for geno, ref in ((g1, ref1), (g2, ref2), (g1, ref1)):
        c.set_genotypes(ua, geno)
        with pytest.raises(mc.HipError):
            c.em_step(0, 0)                                 # a new data set drops the model (its buffers are only parked)
        c.set_model(K, lower_bound=lb)
        assert np.all(c.get_q(0) == 0) and np.all(c.get_p(0) == 0)                    # revived buffers are zeroed like new ones
        c.set_q(0, q0)
        c.set_p(0, p0)
        got = [c.em_step(0, 0) for _ in range(3)] + [c.loglik(0)], c.get_q(0), c.get_p(0), c.expected_counts(), c.data_counts()
        assert same(got, ref)
    # the same data, another model: the parked buffers do not fit and are replaced
    c.set_genotypes(ua, g1)
    c.set_model(K + 2, lower_bound=lb)
    q7, p7 = random_params(I, ua, K + 2, seed=4, lower_bound=lb)
    c.set_q(0, q7)
    c.set_p(0, p7)
    got = [c.em_step(0, 0) for _ in range(3)] + [c.loglik(0)], c.get_q(0), c.get_p(0), c.expected_counts(), c.data_counts()
    assert same(got, fresh_result(ua, g1, K + 2, q7, p7, lb))
    # another shape: everything is replaced
    ua3, g3 = make_dataset(I + 9, L - 7, K, ploidy=4, max_alleles=3, seed=9)
    lb3 = ob.lib.mco_lower_bound(1e-8, I + 9, 4)
    q3, p3 = random_params(I + 9, ua3, K, seed=6, lower_bound=lb3)
    c.set_genotypes(ua3, g3)
    c.set_model(K, lower_bound=lb3)
    c.set_q(0, q3)
    c.set_p(0, p3)
    got = [c.em_step(0, 0) for _ in range(3)] + [c.loglik(0)], c.get_q(0), c.get_p(0), c.expected_counts(), c.data_counts()
    assert same(got, fresh_result(ua3, g3, K, q3, p3, lb3))
    c.close()


@pytest.mark.parametrize("ploidy,missing", [(2, 0.0), (4, 0.02), (3, 0.0)])
def test_copy_genotypes_between_contexts(ploidy, missing):
    I, L, K = 150, 420, 4
    ua, geno = make_dataset(I, L, K, ploidy=ploidy, max_alleles=4, seed=11, missing=missing)
    lb = ob.lib.mco_lower_bound(1e-8, I, ploidy)
    q0, p0 = random_params(I, ua, K, seed=3, lower_bound=lb)
    ref = fresh_result(ua, geno, K, q0, p0, lb)
    src, dst = mc.Context(0), mc.Context(0)
    src.set_genotypes(ua, geno)
    assert src.data_counts() == ref[4]
    cells, copies = ref[4]
    assert copies == int((geno != 255).sum()) and I * L - int((geno == 255).all(axis=2).sum()) <= cells <= copies
    for _ in range(2):                                       # into an empty context, then into one that holds the shape already
        dst.copy_genotypes(src)
        assert np.array_equal(dst.get_genotypes(), geno)
        dst.set_model(K, lower_bound=lb)
        dst.set_q(0, q0)
        dst.set_p(0, p0)
        got = [dst.em_step(0, 0) for _ in range(3)] + [dst.loglik(0)], dst.get_q(0), dst.get_p(0), dst.expected_counts(), dst.data_counts()
        assert same(got, ref)
    src.close()
    dst.close()
