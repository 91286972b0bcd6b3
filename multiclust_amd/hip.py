"""ctypes view of include/multiclust_hip.h (libmulticlust_hip.so)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PROF_KINDS = 4
ABI_VERSION = 2          # MCHIP_ABI_VERSION of include/multiclust_hip.h these bindings were written against
STATUS = {0: "OK", 1: "INVALID", 2: "NO_DEVICE", 3: "HIP", 4: "ALLOC", 5: "STATE", 6: "UNSUPPORTED"}


class HipError(RuntimeError):
    pass


class RunState(C.Structure):
    """mchip_run_state (include/multiclust_hip.h)"""
    _fields_ = [("logL", C.c_double), ("bad_loglik", C.c_double), ("abs_error", C.c_double), ("rel_error", C.c_double),
                ("n_iter", C.c_int), ("max_iter", C.c_int), ("stopped", C.c_int), ("converged", C.c_int),
                ("iter_stop", C.c_int), ("fatal", C.c_int)]


def lib_path():
    # MCHIP_LIB_PATH: an experimental build of the library (scripts/diag/*_exp.sh); the product path is the in-tree one
    return os.environ.get("MCHIP_LIB_PATH") or os.path.join(_HERE, "lib", "libmulticlust_hip.so")


_lib = None


def load():
    """Load the HIP library. Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise HipError("%s is missing: run `make` (or __graft_entry__.build()); there is no CPU fallback" % path)
    lib = C.CDLL(path)
    vp, dp, i32 = C.c_void_p, C.POINTER(C.c_double), C.c_int
    ip = C.POINTER(C.c_int)
    sig = {
        "mchip_abi_version": ([], i32),
        "mchip_device_count": ([ip], i32),
        "mchip_create": ([C.POINTER(vp), i32], i32),
        "mchip_destroy": ([vp], i32),
        "mchip_last_error": ([vp], C.c_char_p),
        "mchip_synchronize": ([vp], i32),
        "mchip_set_genotypes": ([vp, i32, i32, i32, vp, vp], i32),
        "mchip_set_model": ([vp, i32, i32, i32, i32, C.c_double, C.c_double, i32], i32),
        "mchip_set_p": ([vp, i32, vp], i32),
        "mchip_get_p": ([vp, i32, vp], i32),
        "mchip_set_q": ([vp, i32, vp], i32),
        "mchip_get_q": ([vp, i32, vp], i32),
        "mchip_q_length": ([vp, ip], i32),
        "mchip_p_length": ([vp, ip], i32),
        "mchip_em_step": ([vp, i32, i32, dp], i32),
        "mchip_last_loglik": ([vp, dp], i32),
        "mchip_em_run": ([vp, i32, i32, vp], i32),
        "mchip_e_step": ([vp, i32, dp], i32),
        "mchip_loglik": ([vp, i32, dp], i32),
        "mchip_loglik_prefetch": ([vp, i32, dp], i32),
        "mchip_accel_run": ([vp, i32, i32, i32, vp], i32),
        "mchip_mstep_from_partition": ([vp, vp, i32], i32),
        "mchip_mstep_from_rand_partition": ([vp, vp, i32], i32),
        "mchip_get_genotypes": ([vp, vp], i32),
        "mchip_data_counts": ([vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)], i32),
        "mchip_empty_individuals": ([vp, C.POINTER(C.c_int)], i32),
        "mchip_copy_genotypes": ([vp, vp], i32),
        "mchip_simulate_genotypes": ([vp, i32, i32, i32, vp, vp, i32, i32, vp, vp], i32),
        "mchip_set_init_genotypes": ([vp, vp], i32),
        "mchip_get_expected_counts": ([vp, vp], i32),
        "mchip_init_from_allele_centers": ([vp, vp, vp, vp, C.c_uint64, i32], i32),
        "mchip_copy_slot": ([vp, i32, i32], i32),
        "mchip_secant": ([vp, i32, i32, i32, i32], i32),
        "mchip_set_secant": ([vp, i32, i32, vp, vp], i32),
        "mchip_get_secant": ([vp, i32, i32, vp, vp], i32),
        "mchip_step_dots": ([vp, i32, dp], i32),
        "mchip_secant_dots": ([vp, i32, i32, dp], i32),
        "mchip_accel_update": ([vp, i32, i32, i32, C.c_double, i32], i32),
        "mchip_multisecant_update": ([vp, i32, i32, i32, i32, ip, dp, dp], i32),
        "mchip_profile_begin": ([vp], i32),
        "mchip_profile_end": ([vp, dp, dp, ip], i32),
        "mchip_device_info": ([vp, C.c_char_p, i32, ip, dp], i32),
        "mchip_comm_create": ([C.POINTER(vp), i32, ip], i32),
        "mchip_comm_all_reduce": ([vp, C.POINTER(dp), i32, i32], i32),
        "mchip_comm_destroy": ([vp], i32),
        "mchip_comm_last_error": ([vp], C.c_char_p),
        "mchip_comm_info": ([vp, ip, ip, C.POINTER(C.c_ulonglong)], i32),
        "mchip_progress_report": ([C.c_char_p, i32, C.POINTER(C.c_ulonglong)], i32),
        "mchip_progress_note": ([C.c_char_p], i32),
    }
    for name, (args, res) in sig.items():
        if os.environ.get("MCHIP_ALLOW_PARTIAL_ABI") == "1" and not hasattr(lib, name):
            continue                   # diagnostics only (scripts/diag/bits.sh: an older build under comparison); MCHIP_LIB_PATH alone
        fn = getattr(lib, name)        # keeps the strict check: AttributeError here = the library does not export its own header
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


ABI_SYMBOLS = [
    "mchip_abi_version", "mchip_device_count", "mchip_create", "mchip_destroy", "mchip_last_error",
    "mchip_synchronize", "mchip_set_genotypes", "mchip_set_model", "mchip_set_p", "mchip_get_p", "mchip_set_q",
    "mchip_get_q", "mchip_q_length", "mchip_p_length", "mchip_em_step", "mchip_em_run", "mchip_accel_run", "mchip_last_loglik", "mchip_e_step",
    "mchip_loglik", "mchip_loglik_prefetch", "mchip_mstep_from_partition", "mchip_mstep_from_rand_partition",
    "mchip_get_genotypes", "mchip_data_counts", "mchip_empty_individuals", "mchip_copy_genotypes", "mchip_simulate_genotypes", "mchip_set_init_genotypes",
    "mchip_get_expected_counts", "mchip_init_from_allele_centers", "mchip_copy_slot", "mchip_secant", "mchip_set_secant", "mchip_get_secant", "mchip_step_dots",
    "mchip_secant_dots", "mchip_accel_update", "mchip_multisecant_update", "mchip_profile_begin",
    "mchip_profile_end", "mchip_device_info", "mchip_comm_create", "mchip_comm_all_reduce", "mchip_comm_destroy",
    "mchip_comm_last_error", "mchip_comm_info", "mchip_progress_report", "mchip_progress_note",
]


class Context:
    """One HIP context = one stream + device buffers for (device, data set, K)."""

    def __init__(self, device=0):
        self.lib = load()
        self.h = C.c_void_p()
        rc = self.lib.mchip_create(C.byref(self.h), device)
        if rc:
            raise HipError("mchip_create failed: %s (no GPU => no product path; nothing falls back to the CPU)" % STATUS.get(rc, rc))
        self.I = self.L = self.ploidy = self.T = self.K = 0
        self.indiv_q = True

    def _chk(self, rc):
        if rc:
            raise HipError("%s: %s" % (STATUS.get(rc, rc), self.lib.mchip_last_error(self.h).decode()))

    def close(self):
        if self.h:
            self.lib.mchip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_genotypes(self, ua, geno):
        geno = np.ascontiguousarray(geno, dtype=np.uint8)
        ua = np.ascontiguousarray(ua, dtype=np.int32)
        I, L, p = geno.shape
        self._chk(self.lib.mchip_set_genotypes(self.h, I, L, p, ua.ctypes.data, geno.ctypes.data))
        self.I, self.L, self.ploidy, self.T = I, L, p, int(ua.sum())

    def data_counts(self):
        """(cells with n_ic > 0, non-missing allele copies) of the data set held, counted on the device"""
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self.lib.mchip_data_counts(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def copy_genotypes(self, src):
        """take the data set another context (same device) holds"""
        self._chk(self.lib.mchip_copy_genotypes(self.h, src.h))
        self.I, self.L, self.ploidy, self.T = src.I, src.L, src.ploidy, src.T

    def get_genotypes(self):
        g = np.empty((self.I, self.L, self.ploidy), dtype=np.uint8)
        self._chk(self.lib.mchip_get_genotypes(self.h, g.ctypes.data))
        return g

    def set_init_genotypes(self, geno):
        """The data set hard-partition initialisations read instead of the current one (None: back to the current one)."""
        if geno is None:
            self._chk(self.lib.mchip_set_init_genotypes(self.h, None))
            return
        g = np.ascontiguousarray(geno, dtype=np.uint8)
        assert g.shape == (self.I, self.L, self.ploidy)
        self._chk(self.lib.mchip_set_init_genotypes(self.h, g.ctypes.data))

    def simulate_genotypes(self, I, L, ploidy, ua, window, K, q, p, eta_constrained=0):
        """Parametric-bootstrap data set drawn on the device (include/multiclust_hip.h); drops the model."""
        ua = np.ascontiguousarray(ua, dtype=np.int32)
        w = np.ascontiguousarray(window, dtype=np.uint32)
        q = np.ascontiguousarray(q, dtype=np.float64)
        p = np.ascontiguousarray(p, dtype=np.float64)
        assert w.size == 31 and ua.size == L and p.size == K * int(ua.sum())
        assert q.size == (K if eta_constrained else I * K)
        self._chk(self.lib.mchip_simulate_genotypes(self.h, I, L, ploidy, ua.ctypes.data, w.ctypes.data, K,
                                                    eta_constrained, q.ctypes.data, p.ctypes.data))
        self.I, self.L, self.ploidy, self.T = I, L, ploidy, int(ua.sum())

    def set_model(self, K, admixture=1, eta_constrained=0, do_projection=1, lower_bound=1e-8, n_secants=1):
        self._chk(self.lib.mchip_set_model(self.h, K, admixture, eta_constrained, do_projection,
                                           lower_bound, lower_bound, n_secants))
        self.K = K
        self.indiv_q = bool(admixture and not eta_constrained)

    def set_p(self, slot, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        assert p.size == self.K * self.T
        self._chk(self.lib.mchip_set_p(self.h, slot, p.ctypes.data))

    def get_p(self, slot):
        p = np.empty((self.K, self.T), dtype=np.float64)
        self._chk(self.lib.mchip_get_p(self.h, slot, p.ctypes.data))
        return p

    def set_q(self, slot, q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        assert q.size == (self.I * self.K if self.indiv_q else self.K)
        self._chk(self.lib.mchip_set_q(self.h, slot, q.ctypes.data))

    def get_q(self, slot):
        q = np.empty((self.I, self.K) if self.indiv_q else (self.K,), dtype=np.float64)
        self._chk(self.lib.mchip_get_q(self.h, slot, q.ctypes.data))
        return q

    def em_step(self, frm=0, to=0, sync=True):
        if not sync:
            self._chk(self.lib.mchip_em_step(self.h, frm, to, None))
            return None
        ll = C.c_double()
        self._chk(self.lib.mchip_em_step(self.h, frm, to, C.byref(ll)))
        return ll.value

    def last_loglik(self):
        ll = C.c_double()
        self._chk(self.lib.mchip_last_loglik(self.h, C.byref(ll)))
        return ll.value

    def e_step(self, slot=0):
        ll = C.c_double()
        self._chk(self.lib.mchip_e_step(self.h, slot, C.byref(ll)))
        return ll.value

    def loglik(self, slot=0, sync=True):
        if not sync:
            self._chk(self.lib.mchip_loglik(self.h, slot, None))
            return None
        ll = C.c_double()
        self._chk(self.lib.mchip_loglik(self.h, slot, C.byref(ll)))
        return ll.value

    def loglik_prefetch(self, slot=0):
        ll = C.c_double()
        self._chk(self.lib.mchip_loglik_prefetch(self.h, slot, C.byref(ll)))
        return ll.value

    def mstep_from_partition(self, assign, to=0):
        a = np.ascontiguousarray(assign, dtype=np.uint8)
        assert a.size == self.I * self.L * self.ploidy
        self._chk(self.lib.mchip_mstep_from_partition(self.h, a.ctypes.data, to))

    def mstep_from_rand_partition(self, window, to=0):
        """window: the 31 uint32 words behind the first draw, oldest first (see include/multiclust_hip.h)."""
        w = np.ascontiguousarray(window, dtype=np.uint32)
        assert w.size == 31
        self._chk(self.lib.mchip_mstep_from_rand_partition(self.h, w.ctypes.data, to))

    def expected_counts(self):
        s = np.empty((self.I, self.K), dtype=np.float64)
        self._chk(self.lib.mchip_get_expected_counts(self.h, s.ctypes.data))
        return s

    def secant(self, which, j, to, frm):
        self._chk(self.lib.mchip_secant(self.h, which, j, to, frm))

    def step_dots(self, j):
        out = (C.c_double * 3)()
        self._chk(self.lib.mchip_step_dots(self.h, j, out))
        return list(out)

    def secant_dots(self, j1, j2):
        out = (C.c_double * 2)()
        self._chk(self.lib.mchip_secant_dots(self.h, j1, j2, out))
        return list(out)

    def accel_update(self, to, base, j, s, qn_form=0):
        self._chk(self.lib.mchip_accel_update(self.h, to, base, j, s, qn_form))

    def synchronize(self):
        self._chk(self.lib.mchip_synchronize(self.h))

    def profile_begin(self):
        self._chk(self.lib.mchip_profile_begin(self.h))

    def profile_end(self):
        total = C.c_double()
        km = (C.c_double * PROF_KINDS)()
        kl = (C.c_int * PROF_KINDS)()
        self._chk(self.lib.mchip_profile_end(self.h, C.byref(total), km, kl))
        return total.value, list(km), list(kl)

    def device_info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int()
        mem = C.c_double()
        self._chk(self.lib.mchip_device_info(self.h, name, 256, C.byref(cu), C.byref(mem)))
        return name.value.decode(), cu.value, mem.value
