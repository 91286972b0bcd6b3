"""Loader for the committed golden vectors (tests/golden/<case>/), produced from the reference by
oracle/make_fixtures.py + oracle/ref_harness.c.  Data only: inputs and the reference's outputs."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ADMIX_CASES = ["c1_admix_k3", "multi_admix_k4", "tetra_admix_k3", "missing_admix_k3"]
ALL_CASES = sorted(d for d in os.listdir(GOLD) if os.path.exists(os.path.join(GOLD, d, "q0.f64")))     # full EM fixtures
RANDEM_CASES = [d for d in ALL_CASES if os.path.exists(os.path.join(GOLD, d, "randem_ll.f64"))]                # Rand-EM dumps
ACCEL_CASES = [d for d in ALL_CASES if os.path.exists(os.path.join(GOLD, d, "accel_trace.f64"))]         # -s 1..6 given


class Golden:
    def __init__(self, name):
        self.name = name
        self.dir = os.path.join(GOLD, name)
        with open(os.path.join(self.dir, "manifest.json")) as f:
            self.m = json.load(f)
        m = self.m
        self.I, self.L, self.ploidy, self.K, self.T = m["I"], m["L"], m["ploidy"], m["K"], m["T"]
        self.ua = self.i32("uniquealleles.i32")
        self.geno = np.fromfile(os.path.join(self.dir, "geno.u8"), dtype=np.uint8).reshape(self.I, self.L, self.ploidy)
        self.indiv_q = bool(m["admixture"] and not m["eta_constrained"])
        self.lower_bound = float.fromhex(m["lower_bound_hex"])

    def has(self, fn):
        return os.path.exists(os.path.join(self.dir, fn))

    def f64(self, fn):
        return np.fromfile(os.path.join(self.dir, fn), dtype=np.float64)

    def i32(self, fn):
        return np.fromfile(os.path.join(self.dir, fn), dtype=np.int32)

    def p(self, tag):
        return self.f64("p_%s.f64" % tag if not tag.startswith("p") else tag + ".f64").reshape(self.K, self.T)

    def q(self, tag):
        a = self.f64("q_%s.f64" % tag if not tag.startswith("q") else tag + ".f64")
        return a.reshape(self.I, self.K) if self.indiv_q else a

    def sik(self, tag):
        return self.f64("sik_%s.f64" % tag).reshape(self.I, self.K)

    def ilm(self):
        return self.i32("ilm.i32").reshape(self.I, self.T)

    def cycle_states(self):
        """{cycle: (q, p)}: the iterate the reference's accelerated run started that cycle from (accel_states.f64)"""
        nq = self.I * self.K if self.indiv_q else self.K
        rows = self.f64("accel_states.f64").reshape(-1, 1 + nq + self.K * self.T)
        return {int(r[0]): (r[1:1 + nq].reshape(self.I, self.K) if self.indiv_q else r[1:1 + nq], r[1 + nq:].reshape(self.K, self.T))
                for r in rows}

    def cycle_secants(self):
        """{cycle: (delta_index, [(u_q, u_p, v_q, v_p) for j in range(q)])}: what a quasi-Newton run with q > 1 secant pairs
        carries into that cycle besides the iterate (accel_secants.f64; model::delta_index and u_/v_ etaik, pklm); {} if q = 1"""
        if not self.has("accel_secants.f64"):
            return {}
        nq, KT, nsec = (self.I * self.K if self.indiv_q else self.K), self.K * self.T, self.m["q"]
        rows = self.f64("accel_secants.f64").reshape(-1, 2 + nsec * 2 * (nq + KT))
        out = {}
        for r in rows:
            parts, x = [], 2
            for _ in range(nsec):
                four = []
                for _w in range(2):
                    four += [r[x:x + nq], r[x + nq:x + nq + KT].reshape(self.K, self.T)]
                    x += nq + KT
                parts.append(tuple(four))
            out[int(r[0])] = (int(r[1]), parts)
        return out


def ulp_diff(a, b):
    """max distance in units-in-the-last-place between two float64 arrays (same sign assumed for large values)."""
    a = np.ascontiguousarray(a, dtype=np.float64).ravel()
    b = np.ascontiguousarray(b, dtype=np.float64).ravel()
    ai = a.view(np.int64).copy()
    bi = b.view(np.int64).copy()
    ai[ai < 0] = np.int64(-2**63) - ai[ai < 0]
    bi[bi < 0] = np.int64(-2**63) - bi[bi < 0]
    return int(np.max(np.abs(ai - bi))) if a.size else 0
