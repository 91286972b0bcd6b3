"""ctypes binding of oracle/libmc_oracle.so (TEST INFRASTRUCTURE: the CPU checker, never the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(ROOT, "oracle", "libmc_oracle.so")


class Options(C.Structure):
    _fields_ = [
        ("admixture", C.c_int), ("eta_constrained", C.c_int), ("do_projection", C.c_int),
        ("accel_scheme", C.c_int), ("n_init_iter", C.c_int), ("max_iter", C.c_int),
        ("adjust_step", C.c_int), ("abs_error", C.c_double), ("rel_error", C.c_double),
        ("lower_bound", C.c_double), ("fused", C.c_int),
    ]


class Rng(C.Structure):
    _fields_ = [("r", C.c_int32 * 31), ("f", C.c_int), ("b", C.c_int)]


class Summary(C.Structure):
    _fields_ = [("n_init", C.c_int), ("n_total_iter", C.c_int), ("n_max_iter", C.c_int), ("n_maxll_times", C.c_int),
                ("n_maxll_init", C.c_int), ("ever_converged", C.c_int), ("best_unit", C.c_int),
                ("max_logL", C.c_double), ("first_max_logL", C.c_double)]


def _load():
    if not os.path.exists(_LIB):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libmc_oracle.so"], check=True)
    lib = C.CDLL(_LIB)
    vp, dp, i32 = C.c_void_p, C.POINTER(C.c_double), C.c_int
    lib.mco_lower_bound.restype = C.c_double
    lib.mco_lower_bound.argtypes = [C.c_double, i32, i32]
    lib.mco_srand.argtypes = [C.POINTER(Rng), C.c_uint]
    lib.mco_rand.argtypes = [C.POINTER(Rng)]
    lib.mco_rand.restype = i32
    lib.mco_data_create.restype = vp
    lib.mco_data_create.argtypes = [i32, i32, i32, vp, vp]
    lib.mco_data_free.argtypes = [vp]
    lib.mco_data_T.argtypes = [vp]
    lib.mco_data_ilm.restype = C.POINTER(C.c_int32)
    lib.mco_data_ilm.argtypes = [vp]
    lib.mco_model_create.restype = vp
    lib.mco_model_create.argtypes = [vp, C.POINTER(Options), i32]
    lib.mco_model_free.argtypes = [vp]
    lib.mco_model_delta_index.argtypes = [C.c_void_p]
    lib.mco_model_set_delta_index.argtypes = [C.c_void_p, C.c_int]
    lib.mco_model_set_delta_index.restype = None
    for fn in ("mco_model_p", "mco_model_q", "mco_model_u_p", "mco_model_v_p", "mco_model_u_q", "mco_model_v_q"):
        getattr(lib, fn).restype = dp
        getattr(lib, fn).argtypes = [vp, i32]
    lib.mco_model_sik.restype = dp
    lib.mco_model_sik.argtypes = [vp]
    for fn in ("mco_model_q_len", "mco_model_n_iter", "mco_model_converged", "mco_model_pindex",
               "mco_model_findex", "mco_model_tindex", "mco_model_fatal"):
        getattr(lib, fn).restype = i32
        getattr(lib, fn).argtypes = [vp]
    lib.mco_model_logL.restype = C.c_double
    lib.mco_model_logL.argtypes = [vp]
    lib.mco_model_reset.argtypes = [vp]
    lib.mco_michelot_project.argtypes = [dp, i32, C.c_double, C.c_double]
    lib.mco_random_initialize_admixture.argtypes = [vp, C.POINTER(Options), vp, C.POINTER(Rng)]
    lib.mco_initialize_from_partition.argtypes = [vp, C.POINTER(Options), vp, vp]
    lib.mco_randem_initialize.argtypes = [vp, C.POINTER(Options), vp, C.POINTER(Rng), i32, dp]
    lib.mco_random_allele_center.argtypes = [vp, i32, C.POINTER(Rng), vp]
    lib.mco_initialize_parameters_admixture.argtypes = [vp, C.POINTER(Options), vp, vp]
    lib.mco_em_e_step.argtypes = [vp, C.POINTER(Options), vp]
    lib.mco_em_e_step.restype = C.c_double
    lib.mco_em_step.argtypes = [vp, C.POINTER(Options), vp]
    lib.mco_em_step.restype = i32
    lib.mco_e_step.argtypes = [vp, C.POINTER(Options), vp]
    lib.mco_e_step.restype = C.c_double
    lib.mco_log_likelihood.argtypes = [vp, C.POINTER(Options), vp, i32]
    lib.mco_log_likelihood.restype = C.c_double
    lib.mco_em_2_steps.argtypes = [vp, C.POINTER(Options), vp]
    lib.mco_step_size.argtypes = [vp, C.POINTER(Options), vp]
    lib.mco_step_size.restype = C.c_double
    lib.mco_accelerated_update.argtypes = [vp, C.POINTER(Options), vp, C.c_double]
    lib.mco_accelerated_update.restype = C.c_double
    lib.mco_accelerated_em_step.argtypes = [vp, C.POINTER(Options), vp, dp]
    lib.mco_accelerated_em_step.restype = i32
    lib.mco_em.argtypes = [vp, C.POINTER(Options), vp]
    lib.mco_summary_reset.argtypes = [C.POINTER(Summary)]
    lib.mco_summary_add.argtypes = [C.POINTER(Options), C.POINTER(Summary), i32, C.c_double, i32, i32, i32]
    lib.mco_maximize_likelihood.argtypes = [vp, C.POINTER(Options), vp, C.POINTER(Rng), i32, dp, C.POINTER(Summary)]
    return lib


lib = _load()


def make_options(admixture=1, eta_constrained=0, do_projection=1, accel_scheme=0, n_init_iter=0,
                 max_iter=0, adjust_step=0, abs_error=1e-4, rel_error=0.0, lower_bound=1e-8, fused=0):
    return Options(admixture, eta_constrained, do_projection, accel_scheme, n_init_iter, max_iter,
                   adjust_step, abs_error, rel_error, lower_bound, fused)


class Data:
    def __init__(self, I, L, ploidy, ua, geno):
        self.I, self.L, self.ploidy = I, L, ploidy
        self.ua = np.ascontiguousarray(ua, dtype=np.int32)
        self.geno = np.ascontiguousarray(geno, dtype=np.uint8).reshape(I, L, ploidy)
        self.h = lib.mco_data_create(I, L, ploidy, self.ua.ctypes.data, self.geno.ctypes.data)
        self.T = lib.mco_data_T(self.h)

    def ilm(self):
        return np.ctypeslib.as_array(lib.mco_data_ilm(self.h), shape=(self.I, self.T)).copy()

    def __del__(self):
        if getattr(self, "h", None):
            lib.mco_data_free(self.h)
            self.h = None


class Model:
    def __init__(self, data, opt, K):
        self.data, self.opt, self.K = data, opt, K
        self.h = lib.mco_model_create(data.h, C.byref(opt), K)
        self.nq = lib.mco_model_q_len(self.h)

    def p(self, slot):
        return np.ctypeslib.as_array(lib.mco_model_p(self.h, slot), shape=(self.K, self.data.T))

    def q(self, slot):
        a = np.ctypeslib.as_array(lib.mco_model_q(self.h, slot), shape=(self.nq,))
        indiv = self.opt.admixture and not self.opt.eta_constrained
        return a.reshape(self.data.I, self.K) if indiv else a

    def sik(self):
        return np.ctypeslib.as_array(lib.mco_model_sik(self.h), shape=(self.data.I, self.K))

    def u_p(self, j):
        return np.ctypeslib.as_array(lib.mco_model_u_p(self.h, j), shape=(self.K, self.data.T))

    def v_p(self, j):
        return np.ctypeslib.as_array(lib.mco_model_v_p(self.h, j), shape=(self.K, self.data.T))

    def u_q(self, j):
        return np.ctypeslib.as_array(lib.mco_model_u_q(self.h, j), shape=(self.nq,))

    def v_q(self, j):
        return np.ctypeslib.as_array(lib.mco_model_v_q(self.h, j), shape=(self.nq,))

    def set_delta_index(self, di):
        lib.mco_model_set_delta_index(self.h, int(di))

    def reset(self):
        lib.mco_model_reset(self.h)

    def init_random(self, seed):
        rng = Rng()
        lib.mco_srand(C.byref(rng), seed)
        lib.mco_random_initialize_admixture(self.data.h, C.byref(self.opt), self.h, C.byref(rng))
        return rng

    def init_randem(self, seed, n_candidates):
        """Rand-EM initialisation from srand(seed); returns (generator afterwards, per-candidate log likelihoods)"""
        rng = Rng()
        lib.mco_srand(C.byref(rng), seed)
        ll = np.zeros(max(1, n_candidates))
        lib.mco_randem_initialize(self.data.h, C.byref(self.opt), self.h, C.byref(rng), n_candidates, ll.ctypes.data_as(C.POINTER(C.c_double)))
        return rng, ll

    def init_from_partition(self, assign):
        a = np.ascontiguousarray(assign, dtype=np.uint8)
        lib.mco_initialize_from_partition(self.data.h, C.byref(self.opt), self.h, a.ctypes.data)

    def em_step(self):
        return lib.mco_em_step(self.data.h, C.byref(self.opt), self.h)

    def e_step(self):
        return lib.mco_e_step(self.data.h, C.byref(self.opt), self.h)

    def loglik(self, slot):
        return lib.mco_log_likelihood(self.data.h, C.byref(self.opt), self.h, slot)

    def accelerated_em_step(self):
        tr = (C.c_double * 4)()
        stop = lib.mco_accelerated_em_step(self.data.h, C.byref(self.opt), self.h, tr)
        return stop, list(tr)

    def em(self):
        lib.mco_em(self.data.h, C.byref(self.opt), self.h)

    def maximize_likelihood(self, seed, n_units):
        rng = Rng()
        lib.mco_srand(C.byref(rng), seed)
        per = np.zeros((n_units, 4))
        s = Summary()
        lib.mco_maximize_likelihood(self.data.h, C.byref(self.opt), self.h, C.byref(rng), n_units,
                                    per.ctypes.data_as(C.POINTER(C.c_double)), C.byref(s))
        return per, s, lib.mco_rand(C.byref(rng))

    def fit_from_rng(self, rng):
        """one initialisation from a given stream state + em() (what one sharded unit does)"""
        lib.mco_model_reset(self.h)
        lib.mco_random_initialize_admixture(self.data.h, C.byref(self.opt), self.h, C.byref(rng))
        lib.mco_em(self.data.h, C.byref(self.opt), self.h)

    logL = property(lambda s: lib.mco_model_logL(s.h))
    n_iter = property(lambda s: lib.mco_model_n_iter(s.h))
    converged = property(lambda s: lib.mco_model_converged(s.h))
    pindex = property(lambda s: lib.mco_model_pindex(s.h))
    findex = property(lambda s: lib.mco_model_findex(s.h))
    tindex = property(lambda s: lib.mco_model_tindex(s.h))
    fatal = property(lambda s: lib.mco_model_fatal(s.h))

    def __del__(self):
        if getattr(self, "h", None):
            lib.mco_model_free(self.h)
            self.h = None


def michelot(x, minimum, total=1.0):
    x = np.array(x, dtype=np.float64)
    lib.mco_michelot_project(x.ctypes.data_as(C.POINTER(C.c_double)), len(x), total, minimum)
    return x


def glibc_window(seed, skip=0):
    """The 31 words behind draw number `skip` of the stream srand(seed) starts, oldest first, and the next n draws'
    generator (for mchip_mstep_from_rand_partition)."""
    rng = Rng()
    lib.mco_srand(C.byref(rng), seed)
    for _ in range(skip):
        lib.mco_rand(C.byref(rng))
    window = np.array([rng.r[(rng.f + t) % 31] for t in range(31)], dtype=np.int64).astype(np.uint32)
    return window, rng


def rand_mod(rng, n, K):
    """n draws of rand() % K from the oracle's generator (a C loop would be faster; n is small in tests)."""
    return np.fromiter((lib.mco_rand(C.byref(rng)) % K for _ in range(n)), dtype=np.uint8, count=n)


def glibc_rand(seed, n):
    rng = Rng()
    lib.mco_srand(C.byref(rng), seed)
    return [lib.mco_rand(C.byref(rng)) for _ in range(n)]
