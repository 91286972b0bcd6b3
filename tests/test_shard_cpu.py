"""CPU tests of the N > 1 path (world_size-2 gloo): unit -> rank map, the libc-compatible stream's jump-ahead,
the single all-reduce exchange and the serial-order bookkeeping replay (multiclust_amd/shard.py,
multiclust_amd/host/mc_fit.c).  The fits themselves need a GPU, so here each rank's units are fitted by the
oracle (as the checker standing in for the device); what is under test is everything around the fits, and the
result must equal the reference's own serial maximize_likelihood() summary (golden)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

import oracle_bind as ob
from golden_util import Golden
from multiclust_amd import host, shard


def test_rng_jump_equals_stepping():
    lib = host.load()
    for seed in (1, 1234567, 42):
        for n in (1, 2, 30, 31, 32, 1000, 123457, 40000 * 3):
            a, b = host.McRng(), host.McRng()
            lib.mc_srand(C.byref(a), seed)
            lib.mc_srand(C.byref(b), seed)
            for _ in range(n):
                lib.mc_rand(C.byref(a))
            lib.mc_rng_jump(C.byref(b), n)
            assert [lib.mc_rand(C.byref(a)) for _ in range(64)] == [lib.mc_rand(C.byref(b)) for _ in range(64)], (seed, n)


def test_rng_jump_large_is_composable():
    lib = host.load()
    a, b = host.McRng(), host.McRng()
    lib.mc_srand(C.byref(a), 7)
    lib.mc_srand(C.byref(b), 7)
    big = 2 * 10**9 * 7 + 12345          # config-3 scale: 7 initialisations of 2e9 draws
    lib.mc_rng_jump(C.byref(a), big)
    for part in (10**9, 10**9 * 13, 12345):
        lib.mc_rng_jump(C.byref(b), part)
    assert [lib.mc_rand(C.byref(a)) for _ in range(40)] == [lib.mc_rand(C.byref(b)) for _ in range(40)]


def test_host_rand_matches_glibc_known_answers():
    lib = host.load()
    g = host.McRng()
    lib.mc_srand(C.byref(g), 1)
    assert [lib.mc_rand(C.byref(g)) for _ in range(3)] == [1804289383, 846930886, 1681692777]


def test_bookkeeping_replay_matches_reference_summary():
    g = Golden("c1_admix_k3")
    ref = g.f64("multi_init.f64").reshape(-1, 4)
    opt = host.McOptions()
    host.load().mc_make_options(C.byref(opt))
    results = [host.McUnitResult(u, ref[u, 0], int(ref[u, 1]), int(ref[u, 2]), 0, 0, int(ref[u, 3]), 0) for u in range(len(ref))]
    s = shard.replay(results, opt, g.m["no_parameters"], g.I)
    for k in ("n_init", "n_total_iter", "n_max_iter", "n_maxll_times", "n_maxll_init", "ever_converged"):
        assert getattr(s, k) == g.m["mi_" + k], k
    assert s.max_logL == g.m["mi_max_logL"] and s.first_max_logL == g.m["mi_first_max_logL"]
    assert abs(s.aic - g.m["mi_aic"]) <= 1e-9 and abs(s.bic - g.m["mi_bic"]) <= 1e-9
    assert s.best_unit == int(np.argmax(ref[:, 0]))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, name, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = Golden(name)
    lib = host.load()
    mopt = host.McOptions()
    lib.mc_make_options(C.byref(mopt))
    mopt.admixture, mopt.accel_scheme = 1, g.m["accel_scheme"]
    ua = np.ascontiguousarray(g.ua)
    geno = np.ascontiguousarray(g.geno)
    mdat = host.McData(g.I, g.L, g.ploidy, ua.ctypes.data, geno.ctypes.data)
    assert lib.mc_synchronize(C.byref(mopt), C.byref(mdat)) == 0
    draws = lib.mc_draws_per_init(C.byref(mopt), C.byref(mdat), g.K)
    n_units = g.m["mi_units"]
    # the stand-in fitter: oracle, reference order
    oopt = ob.make_options(admixture=1, accel_scheme=g.m["accel_scheme"], lower_bound=mopt.lower_bound,
                           abs_error=mopt.abs_error, rel_error=mopt.rel_error)
    mod = ob.Model(ob.Data(g.I, g.L, g.ploidy, g.ua, g.geno), oopt, g.K)
    mine = []
    for u in shard.units_for_rank(n_units, rank, world):
        rng = host.McRng()
        lib.mc_srand(C.byref(rng), g.m["seed"])
        lib.mc_rng_jump(C.byref(rng), u * draws)          # product code: where the serial program would be
        orng = ob.Rng()
        C.memmove(C.byref(orng), C.byref(rng), C.sizeof(rng))
        mod.fit_from_rng(orng)
        mine.append(host.McUnitResult(u, mod.logL, mod.converged, mod.n_iter, 0, 0, mod.pindex, mod.fatal))
    allres = shard.exchange(mine, n_units, dist)          # the path's single collective
    s = shard.replay(allres, mopt, g.m["no_parameters"], g.I)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank),
            np.array([[r.logL, r.converged, r.n_iter, r.pindex] for r in allres] +
                     [[s.n_init, s.n_total_iter, s.n_maxll_times, s.best_unit], [s.max_logL, s.first_max_logL, s.n_maxll_init, s.n_max_iter]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["multi_admix_k4", "missing_admix_k3"])
def test_two_rank_gloo_sharded_initialisations_reproduce_serial_reference(name, tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_rank_main, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    g = Golden(name)
    ref = g.f64("multi_init.f64").reshape(-1, 4)
    outs = [np.load(tmp_path / ("rank%d.npy" % r)) for r in range(world)]
    assert np.array_equal(outs[0], outs[1])               # every rank holds the same picture
    per, tail = outs[0][:-2], outs[0][-2:]
    assert np.array_equal(per, ref)                       # unit u started exactly where the serial stream would be
    assert list(tail[0][:3]) == [g.m["mi_n_init"], g.m["mi_n_total_iter"], g.m["mi_n_maxll_times"]]
    assert tail[0][3] == int(np.argmax(ref[:, 0])) and shard.owner_of(int(tail[0][3]), world) in (0, 1)
    assert tail[1][0] == g.m["mi_max_logL"] and tail[1][1] == g.m["mi_first_max_logL"]
    assert tail[1][2] == g.m["mi_n_maxll_init"] and tail[1][3] == g.m["mi_n_max_iter"]


def test_threaded_partition_draw_equals_serial_stream():
    """mc_initialize_model draws large partitions on several host threads, each starting from the jumped-ahead
    stream: byte-identical to the serial rand() % K sequence, and the caller's stream ends at the same position."""
    lib = host.load()
    lib.mc_test_draw_partition.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(host.McRng)]
    n, K = (1 << 24) + 12345, 7
    os.environ["MC_INIT_THREADS"] = "5"
    a = np.empty(n, dtype=np.uint8)
    g = host.McRng()
    lib.mc_srand(C.byref(g), 99)
    lib.mc_test_draw_partition(a.ctypes.data, n, K, C.byref(g))
    del os.environ["MC_INIT_THREADS"]
    g2 = host.McRng()
    lib.mc_srand(C.byref(g2), 99)
    ref = np.array([lib.mc_rand(C.byref(g2)) % K for _ in range(200000)], dtype=np.uint8)
    assert np.array_equal(a[:200000], ref)
    lib.mc_rng_jump(C.byref(g2), n - 200000)
    assert [lib.mc_rand(C.byref(g)) for _ in range(8)] == [lib.mc_rand(C.byref(g2)) for _ in range(8)]
    # a slice in the middle of another thread's range
    g3 = host.McRng()
    lib.mc_srand(C.byref(g3), 99)
    lo = 3 * ((n + 4) // 5) + 17
    lib.mc_rng_jump(C.byref(g3), lo)
    assert np.array_equal(a[lo:lo + 1000], np.array([lib.mc_rand(C.byref(g3)) % K for _ in range(1000)], dtype=np.uint8))
