"""Seeded synthetic genotype matrices of the shape SURVEY.md section 8d prescribes:
P_kl. ~ Dirichlet(0.5), Q_i ~ Dirichlet(0.2), each allele copy z ~ Cat(Q_i), allele ~ Cat(P_z,l)."""
import numpy as np


def make_dataset(I, L, K, ploidy=2, max_alleles=2, seed=0, missing=0.0, chunk=2048):
    rng = np.random.default_rng(seed)
    ua = rng.integers(2, max_alleles + 1, size=L).astype(np.int32) if max_alleles > 2 else np.full(L, 2, np.int32)
    M = int(ua.max())
    # cumulative allele distribution per (k, l), padded to M
    P = rng.gamma(0.5, size=(K, L, M)) + 1e-3
    P *= (np.arange(M)[None, None, :] < ua[None, :, None])
    P /= P.sum(axis=2, keepdims=True)
    cdf = np.cumsum(P, axis=2)
    Q = rng.dirichlet(np.full(K, 0.2), size=I)
    qcdf = np.cumsum(Q, axis=1)
    geno = np.empty((I, L, ploidy), dtype=np.uint8)
    lidx = np.arange(L)
    for i0 in range(0, I, chunk):
        i1 = min(I, i0 + chunk)
        n = i1 - i0
        if K > 1:
            z = (rng.random((n, L, ploidy))[..., None] > qcdf[i0:i1, None, None, :-1]).sum(axis=3)
        else:
            z = np.zeros((n, L, ploidy), dtype=np.int64)
        u = rng.random((n, L, ploidy))
        c = cdf[z, lidx[None, :, None], :]                      # (n, L, ploidy, M)
        a = (u[..., None] > c[..., :-1]).sum(axis=3)
        a = np.minimum(a, ua[None, :, None] - 1)
        geno[i0:i1] = a.astype(np.uint8)
    if missing > 0:
        geno[rng.random(geno.shape) < missing] = 0xFF
    return ua, geno


def random_params(I, ua, K, seed=1, lower_bound=1e-8):
    rng = np.random.default_rng(seed)
    T = int(ua.sum())
    q = rng.dirichlet(np.ones(K), size=I)
    p = np.empty((K, T))
    off = 0
    for M in ua:
        p[:, off:off + M] = rng.dirichlet(np.ones(M), size=K)
        off += M
    return np.maximum(q, lower_bound), np.maximum(p, lower_bound)
