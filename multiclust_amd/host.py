"""ctypes view of multiclust_amd/host/mc_host.h (libmulticlust_host.so): the plain-C host side that mirrors
the reference's em()/em_step()/stop()/accelerated_em_step() over the C-ABI."""
import ctypes as C
import os

import numpy as np

from . import hip

_HERE = os.path.dirname(os.path.abspath(__file__))


class McOptions(C.Structure):
    _fields_ = [("admixture", C.c_int), ("eta_constrained", C.c_int), ("do_projection", C.c_int),
                ("accel_scheme", C.c_int), ("q", C.c_int), ("n_init_iter", C.c_int), ("max_iter", C.c_int),
                ("n_seconds", C.c_uint), ("adjust_step", C.c_int), ("verbosity", C.c_int),
                ("abs_error", C.c_double), ("rel_error", C.c_double), ("lower_bound", C.c_double),
                ("eta_lower_bound", C.c_double), ("p_lower_bound", C.c_double), ("seed", C.c_uint),
                ("initialization_procedure", C.c_int), ("n_rand_em_init", C.c_int)]


class McData(C.Structure):
    _fields_ = [("I", C.c_int), ("L", C.c_int), ("ploidy", C.c_int),
                ("uniquealleles", C.c_void_p), ("geno", C.c_void_p), ("init_geno", C.c_void_p)]


class McModel(C.Structure):
    _fields_ = [("K", C.c_int), ("pindex", C.c_int), ("findex", C.c_int), ("tindex", C.c_int),
                ("delta_index", C.c_int), ("logL", C.c_double), ("n_iter", C.c_int), ("converged", C.c_int),
                ("stopped", C.c_int), ("accel_step", C.c_int), ("iter_stop", C.c_int), ("time_stop", C.c_int),
                ("fatal", C.c_int), ("start", C.c_long), ("seconds_run", C.c_double),
                ("A", C.c_double * 9), ("Ainv", C.c_double * 9), ("cutu", C.c_double * 3),
                ("last_emll", C.c_double), ("last_step", C.c_double), ("last_ll", C.c_double),
                ("last_accepted", C.c_int), ("dev", C.c_void_p), ("owns_dev", C.c_int), ("init_cache", C.c_void_p)]


class McRng(C.Structure):
    _fields_ = [("r", C.c_int32 * 31), ("f", C.c_int), ("b", C.c_int)]


class McSimulation(C.Structure):
    _fields_ = [("window", C.c_uint32 * 31), ("K", C.c_int), ("q", C.c_void_p), ("p", C.c_void_p)]


class McUnitResult(C.Structure):
    _fields_ = [("unit", C.c_int), ("logL", C.c_double), ("converged", C.c_int), ("n_iter", C.c_int),
                ("time_stop", C.c_int), ("iter_stop", C.c_int), ("pindex", C.c_int), ("fatal", C.c_int),
                ("seconds_run", C.c_double)]


class McReplicateResult(C.Structure):
    _fields_ = [("replicate", C.c_int), ("logL_H0", C.c_double), ("logL_HA", C.c_double), ("ts", C.c_double),
                ("n_iter", C.c_int), ("fatal", C.c_int)]


class McSummary(C.Structure):
    _fields_ = [("n_init", C.c_int), ("n_total_iter", C.c_int), ("n_max_iter", C.c_int),
                ("n_maxll_times", C.c_int), ("n_maxll_init", C.c_int), ("ever_converged", C.c_int),
                ("best_unit", C.c_int), ("max_logL", C.c_double), ("first_max_logL", C.c_double),
                ("aic", C.c_double), ("bic", C.c_double)]


class CliOptions(C.Structure):
    """mc_cli_options (multiclust_amd/host/mc_cli.h)"""
    _fields_ = [("em", McOptions), ("filename", C.c_char_p), ("filename_file", C.c_char_p), ("path", C.c_char_p),
                ("outfile_name", C.c_char_p), ("min_K", C.c_int), ("max_K", C.c_int), ("n_init", C.c_int),
                ("n_bootstrap", C.c_int), ("n_rand_em_init", C.c_int), ("missing_value", C.c_int), ("R_format", C.c_int),
                ("ploidy", C.c_int), ("seed_given", C.c_int), ("target_ll", C.c_int), ("target_revisit", C.c_int),
                ("desired_ll", C.c_double), ("n_repeat", C.c_int), ("repeat_seconds", C.c_uint),
                ("max_repeat_seconds", C.c_uint), ("write_files", C.c_int), ("compact", C.c_int), ("parallel", C.c_int),
                ("device", C.c_int), ("n_gpus", C.c_int), ("n_streams", C.c_int), ("pfile", C.c_char_p), ("qfile", C.c_char_p),
                ("afile", C.c_char_p)]


class CliData(C.Structure):
    """mc_cli_data (multiclust_amd/host/mc_cli.h)"""
    _fields_ = [("I", C.c_int), ("L", C.c_int), ("ploidy", C.c_int), ("M", C.c_int), ("missing_data", C.c_int),
                ("interleaved", C.c_int), ("IL", C.POINTER(C.c_int)), ("uniquealleles", C.POINTER(C.c_int32)),
                ("L_alleles", C.POINTER(C.POINTER(C.c_int))), ("geno", C.POINTER(C.c_uint8)),
                ("names", C.POINTER(C.c_char_p)), ("locale", C.POINTER(C.c_int)), ("pops", C.POINTER(C.c_char_p)),
                ("numpops", C.c_int), ("i_p", C.POINTER(C.c_int)), ("T", C.c_int), ("toff", C.POINTER(C.c_int32))]


def read_structure(path, ploidy=2, missing=-9, r_format=0):
    """mc_read_structure (host/mc_reader.c; reference read_file.c:38-300, 443-663) on a STRUCTURE file: (status, None) on
    failure, else (0, dict of the fields the EM path and the writers read)."""
    lib = load()
    lib.mc_read_structure.argtypes = [C.POINTER(CliOptions), C.POINTER(CliData)]
    lib.mc_free_data.argtypes = [C.POINTER(CliData)]
    o = CliOptions()
    o.filename = path.encode()
    o.ploidy, o.missing_value, o.R_format = ploidy, missing, r_format
    d = CliData()
    rc = lib.mc_read_structure(C.byref(o), C.byref(d))
    if rc:
        return rc, None
    out = dict(I=d.I, L=d.L, ploidy=d.ploidy, T=d.T, M=d.M, missing_data=d.missing_data, interleaved=d.interleaved,
               ua=np.ctypeslib.as_array(d.uniquealleles, shape=(d.L,)).copy(),
               geno=np.ctypeslib.as_array(d.geno, shape=(d.I, d.L, d.ploidy)).copy(),
               locale=np.ctypeslib.as_array(d.locale, shape=(d.I,)).copy(), numpops=d.numpops,
               names=[d.names[i].decode() for i in range(d.I)], pops=[d.pops[i].decode() for i in range(d.numpops)],
               i_p=[d.i_p[n] for n in range(d.numpops)])
    lib.mc_free_data(C.byref(d))
    return 0, out


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    hip.load()      # dependency; raises loudly if the HIP library is missing
    path = os.path.join(_HERE, "lib", "libmulticlust_host.so")
    if not os.path.exists(path):
        raise hip.HipError("%s is missing: run `make`" % path)
    lib = C.CDLL(path)
    OP, DP, MP = C.POINTER(McOptions), C.POINTER(McData), C.POINTER(McModel)
    lib.mc_make_options.argtypes = [OP]
    lib.mc_synchronize.argtypes = [OP, DP]
    lib.mc_model_create.argtypes = [C.POINTER(MP), OP, DP, C.c_int, C.c_int]
    lib.mc_model_free.argtypes = [MP]
    for fn in ("mc_model_set_p", "mc_model_get_p", "mc_model_set_q", "mc_model_get_q"):
        getattr(lib, fn).argtypes = [MP, C.c_int, C.c_void_p]
    lib.mc_model_get_expected_counts.argtypes = [MP, C.c_void_p]
    lib.mc_initialize_model.argtypes = [OP, DP, MP, C.POINTER(McRng)]
    lib.mc_reset_model_state.argtypes = [MP]
    lib.mc_skip_initializations.argtypes = [OP, DP, MP, C.POINTER(McRng), C.c_int]
    lib.mc_em.argtypes = [OP, DP, MP]
    lib.mc_em.restype = None
    lib.mc_em_step.argtypes = [OP, DP, MP]
    lib.mc_em_e_step.argtypes = [OP, DP, MP]
    lib.mc_em_e_step.restype = C.c_double
    lib.mc_em_2_steps.argtypes = [MP, DP, OP]
    lib.mc_accelerated_em_step.argtypes = [OP, DP, MP]
    lib.mc_log_likelihood.argtypes = [OP, DP, MP, C.c_int]
    lib.mc_log_likelihood.restype = C.c_double
    lib.mc_step_size.argtypes = [OP, DP, MP]
    lib.mc_step_size.restype = C.c_double
    lib.mc_accelerated_update.argtypes = [OP, DP, MP, C.c_double]
    lib.mc_accelerated_update.restype = C.c_double
    lib.mc_srand.argtypes = [C.POINTER(McRng), C.c_uint]
    lib.mc_rand.argtypes = [C.POINTER(McRng)]
    lib.mc_rng_jump.argtypes = [C.POINTER(McRng), C.c_uint64]
    lib.mc_summary_reset.argtypes = [C.POINTER(McSummary)]
    lib.mc_summary_add.argtypes = [OP, C.POINTER(McSummary), C.POINTER(McUnitResult), C.c_int, C.c_int]
    lib.mc_draws_per_init.argtypes = [OP, DP, C.c_int]
    lib.mc_draws_per_init.restype = C.c_uint64
    lib.mc_fit_unit.argtypes = [OP, DP, MP, C.c_uint, C.c_int, C.POINTER(McUnitResult)]
    lib.mc_no_parameters.argtypes = [OP, DP, C.c_int]
    lib.mc_bootstrap_genotypes.argtypes = [OP, DP, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(McRng), C.c_void_p]
    lib.mc_bootstrap_genotypes.restype = None
    lib.mc_bootstrap_draws.argtypes = [OP, DP]
    lib.mc_bootstrap_draws.restype = C.c_uint64
    lib.mc_simulation_begin.argtypes = [C.POINTER(McSimulation), OP, DP, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(McRng)]
    lib.mc_simulation_begin.restype = None
    lib.mc_model_create_simulated.argtypes = [C.POINTER(MP), OP, DP, C.c_int, C.c_int, C.POINTER(McSimulation)]
    lib.mc_model_get_genotypes.argtypes = [MP, C.c_void_p]
    lib.mc_fit_replicate.argtypes = [OP, DP, C.c_int, C.POINTER(McRng), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.POINTER(McReplicateResult), C.POINTER(MP)]
    lib.mc_aic.restype = C.c_double
    lib.mc_aic.argtypes = [C.c_double, C.c_int]
    lib.mc_bic.restype = C.c_double
    lib.mc_bic.argtypes = [C.c_double, C.c_int, C.c_int]
    _lib = lib
    return lib


class Fit:
    """options + data + model triple, the argument convention of the reference's EM layer."""

    def __init__(self, ua, geno, K, device=0, **opts):
        self.lib = load()
        self.ua = np.ascontiguousarray(ua, dtype=np.int32)
        self.geno = np.ascontiguousarray(geno, dtype=np.uint8)
        I, L, p = self.geno.shape
        self.opt = McOptions()
        self.lib.mc_make_options(C.byref(self.opt))
        for k, v in opts.items():
            setattr(self.opt, k, v)
        self.dat = McData(I, L, p, self.ua.ctypes.data, self.geno.ctypes.data)
        if self.lib.mc_synchronize(C.byref(self.opt), C.byref(self.dat)):
            raise hip.HipError("mc_synchronize failed")
        self.mp = C.POINTER(McModel)()
        rc = self.lib.mc_model_create(C.byref(self.mp), C.byref(self.opt), C.byref(self.dat), K, device)
        if rc:
            raise hip.HipError("mc_model_create failed with status %d (no GPU => no product path)" % rc)
        self.K, self.I, self.T = K, I, int(self.ua.sum())
        self.indiv_q = bool(self.opt.admixture and not self.opt.eta_constrained)

    mod = property(lambda s: s.mp.contents)

    def close(self):
        if self.mp:
            self.lib.mc_model_free(self.mp)
            self.mp = C.POINTER(McModel)()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _a(self):
        return C.byref(self.opt), C.byref(self.dat), self.mp

    def set_params(self, q, p, slot=0):
        q = np.ascontiguousarray(q, dtype=np.float64)
        p = np.ascontiguousarray(p, dtype=np.float64)
        assert not self.lib.mc_model_set_q(self.mp, slot, q.ctypes.data)
        assert not self.lib.mc_model_set_p(self.mp, slot, p.ctypes.data)

    def get_q(self, slot):
        q = np.empty((self.I, self.K) if self.indiv_q else (self.K,))
        assert not self.lib.mc_model_get_q(self.mp, slot, q.ctypes.data)
        return q

    def get_p(self, slot):
        p = np.empty((self.K, self.T))
        assert not self.lib.mc_model_get_p(self.mp, slot, p.ctypes.data)
        return p

    def expected_counts(self):
        s = np.empty((self.I, self.K))
        assert not self.lib.mc_model_get_expected_counts(self.mp, s.ctypes.data)
        return s

    def reset(self):
        self.lib.mc_reset_model_state(self.mp)

    def initialize(self, seed):
        rng = McRng()
        self.lib.mc_srand(C.byref(rng), seed)
        rc = self.lib.mc_initialize_model(C.byref(self.opt), C.byref(self.dat), self.mp, C.byref(rng))
        if rc:
            raise hip.HipError("mc_initialize_model failed (%d)" % rc)
        return rng

    def em(self):
        self.lib.mc_em(*self._a())

    def em_step(self):
        return self.lib.mc_em_step(*self._a())

    def em_e_step(self):
        return self.lib.mc_em_e_step(*self._a())

    def accelerated_em_step(self):
        return self.lib.mc_accelerated_em_step(*self._a())

    def fit_unit(self, seed, unit):
        """initialisation `unit` of the run seeded with `seed` (stream jumped to the serial program's offset) + em()"""
        r = McUnitResult()
        rc = self.lib.mc_fit_unit(C.byref(self.opt), C.byref(self.dat), self.mp, seed, unit, C.byref(r))
        if rc:
            raise hip.HipError("mc_fit_unit failed (%d)" % rc)
        return r

    def no_parameters(self):
        return self.lib.mc_no_parameters(C.byref(self.opt), C.byref(self.dat), self.K)

    def log_likelihood(self, which):
        return self.lib.mc_log_likelihood(C.byref(self.opt), C.byref(self.dat), self.mp, which)
