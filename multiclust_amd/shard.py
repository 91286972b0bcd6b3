"""Sharding of independent EM fits (random initialisations, bootstrap replicates) over ranks.

The path shards by units (reference multiclust.c:516-653, 681-701): unit u runs on rank u % world with no
data-path collective.  One all-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests) makes every rank hold every
unit's scalars; each rank then replays the serial program's bookkeeping in unit order
(multiclust_amd/host/mc_fit.c: mc_summary_add) and so agrees on the winner, whose owner holds the parameters.
"""
import ctypes as C

from . import host

FIELDS = ("logL", "converged", "n_iter", "time_stop", "iter_stop", "pindex", "fatal", "filled")


def units_for_rank(n_units, rank, world):
    return list(range(rank, n_units, world))


def owner_of(unit, world):
    return unit % world


def exchange(local_results, n_units, dist=None, device="cpu"):
    """local_results: iterable of McUnitResult (or objects with the same fields) for this rank's units.
    Returns the list of all n_units results in unit order, identical on every rank (one all-reduce).
    Without a process group (one rank, nothing to exchange) PyTorch is not touched: a process that loads it AFTER the HIP
    library ends up with two HIP runtimes (PyTorch asks for "libamdhip64.so", ROCm's SONAME is "libamdhip64.so.7")."""
    rows = [[0.0] * len(FIELDS) for _ in range(n_units)]
    for r in local_results:
        rows[r.unit] = [float(r.logL), float(r.converged), float(r.n_iter), float(r.time_stop), float(r.iter_stop), float(r.pindex),
                        float(r.fatal), 1.0]
    if dist is not None and dist.is_initialized():    # also with one rank: the rehearsal of the exchange on a one-GPU box
        import torch
        t = torch.tensor(rows, dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)      # rows are disjoint across ranks
        rows = t.cpu().tolist()
    out = []
    for u, row in enumerate(rows):
        if row[7] != 1.0:
            raise RuntimeError("unit %d was fitted by %d ranks" % (u, int(row[7])))
        out.append(host.McUnitResult(u, row[0], int(row[1]), int(row[2]), int(row[3]), int(row[4]), int(row[5]), int(row[6])))
    return out


def replay(results, opt, no_parameters, I):
    """The serial program's summary (n_init, n_total_iter, n_maxll_times, max_logL, best unit, AIC, BIC ...)."""
    lib = host.load()
    s = host.McSummary()
    lib.mc_summary_reset(C.byref(s))
    for r in results:
        lib.mc_summary_add(C.byref(opt), C.byref(s), C.byref(r), no_parameters, I)
    return s
