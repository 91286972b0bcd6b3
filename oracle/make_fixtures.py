#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: generate the committed golden vectors under tests/golden/.

Runs ONLY in the build container (needs /root/reference and oracle/_ref/ref_harness, built by
`make -C oracle ref`).  For each fixture it (1) writes a small synthetic STRUCTURE-format input with a
fixed Python `random` seed, (2) runs the reference-linked harness (oracle/ref_harness.c) on it, which
dumps inputs and the reference's outputs at full precision.  Fixtures are data only (inputs + expected
outputs); no reference source is copied.  Re-run: `python oracle/make_fixtures.py`.
"""
import json, os, random, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


def write_stru(path, I, L, K, ploidy, nalleles, seed, missing=0.0, allele_codes=None):
    """nalleles: list of per-locus allele counts. Individuals are admixed 0.8/0.2 around locale i%K."""
    rnd = random.Random(seed)
    P = []
    for l in range(L):
        M = nalleles[l]
        rows = []
        for k in range(K):
            g = [rnd.gammavariate(0.5, 1.0) + 1e-3 for _ in range(M)]
            s = sum(g)
            rows.append([x / s for x in g])
        P.append(rows)
    with open(path, "w") as f:
        f.write(" ".join("loc%d" % (l + 1) for l in range(L)) + "\n")
        for i in range(I):
            pop = i % K
            q = [0.2 / (K - 1) if K > 1 else 1.0] * K
            q[pop] = 0.8 if K > 1 else 1.0
            for a in range(ploidy):
                row = []
                for l in range(L):
                    k = rnd.choices(range(K), q)[0]
                    m = rnd.choices(range(nalleles[l]), P[l][k])[0]
                    code = allele_codes[l][m] if allele_codes else m + 1
                    if missing and rnd.random() < missing:
                        code = -9
                    row.append(str(code))
                f.write("ind%d pop%d " % (i, pop) + " ".join(row) + "\n")


def write_c1(path):
    """SURVEY.md App. D generator for the config-1 plumbing case (100 x 500 biallelic, K=3)."""
    random.seed(42)
    I, L, K = 100, 500, 3
    P = [[random.betavariate(0.5, 0.5) for _ in range(L)] for _ in range(K)]
    with open(path, "w") as f:
        f.write(" ".join("loc%d" % (l + 1) for l in range(L)) + "\n")
        for i in range(I):
            pop = i % K
            q = [0.1] * K
            q[pop] = 0.8
            for a in range(2):
                row = []
                for l in range(L):
                    k = random.choices(range(K), q)[0]
                    row.append("1" if random.random() < P[k][l] else "2")
                f.write("ind%d pop%d " % (i, pop) + " ".join(row) + "\n")


BOOTSTRAP_CASES = ("multi_admix_k4", "tetra_admix_k3", "missing_admix_k3", "multi_mix_k3", "multi_admix_c_k3")
# section 6b as well: EM steps on the simulated data set (projection off: alleles the replicate lacks keep p = 0)
BOOTSTRAP_EM_CASES = ("multi_admix_k4_noproj", "rare_admix_k3_noproj", "rare_admix_k3_bs")
# section 7: Rand-EM initialisation with this many candidates
# section 3b also records the iterate every accelerated cycle starts from (stride 1 = every cycle; n: cycles c, c+1 of every n)
CYCLE_STATE_CASES = {"c1_admix_k3": 6, "multi_admix_k4": 1, "multi_admix_k4_s1": 1, "multi_admix_k4_s2": 1,
                     "multi_admix_k3_qn1": 1, "tetra_admix_k3": 1, "missing_admix_k3": 1, "multi_admix_c_k3": 1,
                     "multi_admix_k4_tinybound": 2, "mixslow_mix_k3_s3": 1, "mixslow_mix_k3_qn1": 1,
                     # quasi-Newton with two and three secant pairs (the pairs and delta_index travel with every state), and SQUAREM
                     # with step back-tracking (-g)
                     "multi_admix_k3_qn2": 1, "multi_admix_k3_qn3": 1, "mixslow_mix_k3_qn2": 1, "mixslow_mix_k3_qn3": 1,
                     "multi_admix_k4_g3": 1, "multi_admix_k4_s1_g2": 1}
RANDEM_CASES = {"multi_admix_k3_randem": 6, "wide_admix_k2_randem": 5, "tetra_admix_k3_randem": 4,
                "missing_admix_k2_randem": 5, "multi_admix_c_k3_randem": 4, "multi_mix_k3_randem": 5}


# Every reference process runs in a fresh directory /tmp/mcfix.XXXXXXXX (fixed name length) on a copy of its input files, with a
# fixed minimal environment: what the reference reads from uninitialised heap memory (the phantom allele slot of a locus with
# missing data, read_file.c:527-533 vs 581-585) then does not depend on where this checkout lies or on
# the caller's environment, and `python oracle/make_fixtures.py` gives the same bytes from any checkout path.
FIXED_ENV = {"PATH": "/usr/bin:/bin", "LC_ALL": "C"}
CLI_RECORD = {}


def scratch_copy(files, attempt=1):
    """attempt n lengthens the directory name by n - 1 characters: the reference's heap layout (and with it what the
    uninitialised phantom slots hold, and whether its free() aborts) depends on the lengths of the path strings it copies"""
    import tempfile
    tmp = tempfile.mkdtemp(prefix="mcfix." + "x" * (attempt - 1), dir="/tmp")
    local = []
    for f in files:
        dst = os.path.join(tmp, os.path.basename(f))
        shutil.copyfile(f, dst)
        local.append(dst)
    return tmp, local


def phantom_slots(stru, ploidy):
    """{locus: index of the phantom allele slot} for loci with missing data in a non-interleaved STRUCTURE file whose missing
    code is -9: the reference counts a slot for MISSING but lists only the real alleles (ascending), so the last slot is
    uninitialised memory"""
    rows = [l.split()[2:] for l in open(stru).read().strip().split("\n")[1:]]
    out = {}
    for l in range(len(rows[0])):
        codes = set(r[l] for r in rows)
        if "-9" in codes:
            out[l] = len(codes) - 1
    return out


def run(name, stru, n_em, snaps, n_cycles, args, keep_ilm=True):
    out = os.path.join(GOLD, name)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    env = dict(FIXED_ENV)
    if name in BOOTSTRAP_CASES:     # section 6 of the harness: one parametric-bootstrap data set (bs_ilm.u8)
        env["REF_HARNESS_BOOTSTRAP"] = "1"
    if name in BOOTSTRAP_EM_CASES:
        env["REF_HARNESS_BOOTSTRAP"] = "2"
    if name in CYCLE_STATE_CASES:
        env["REF_HARNESS_CYCLE_STATES"] = str(CYCLE_STATE_CASES[name])
    if name in RANDEM_CASES:
        env["REF_HARNESS_RANDEM"] = str(RANDEM_CASES[name])
    for attempt in range(1, 9):
        # the harness itself refuses a run in which heap garbage matched an allele code ("phantom allele slot has counts"):
        # try again from a scratch directory with a longer name (another heap layout)
        tmp, (local,) = scratch_copy([stru], attempt)
        work = os.path.join(tmp, "o")
        os.makedirs(work)
        cmd = [HARNESS, work, str(n_em), snaps, str(n_cycles), "--", "-f", local] + args
        print(" ".join(cmd))
        res = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, cwd=work, env=env, text=True)
        if res.returncode == 0:
            break
        shutil.rmtree(tmp)
        if "phantom allele slot has counts" not in res.stderr:
            sys.exit("%s: the harness failed: %s" % (name, res.stderr[-2000:]))
    else:
        sys.exit("%s: heap garbage matched an allele code in every attempt" % name)
    for fn in os.listdir(work):
        shutil.copyfile(os.path.join(work, fn), os.path.join(out, fn))
    shutil.rmtree(tmp)
    with open(os.path.join(out, "manifest.json")) as f:
        json.load(f)                # the reference exit(0)s on NaN / decrease: a truncated manifest is a failed fixture
    with open(os.path.join(out, "ARGS.txt"), "w") as f:
        f.write(" ".join(["-f", os.path.basename(stru)] + args) + "\n")
    if not keep_ilm:
        os.remove(os.path.join(out, "ilm.i32"))
    for fn in os.listdir(out):      # stray files the reference may write into cwd
        if fn.endswith(".txt") and fn not in ("ARGS.txt",):
            os.remove(os.path.join(out, fn))


REFBIN = os.path.join(ROOT, "oracle", "_ref", "multiclust_ref")


def run_cli(name, stru, args, ploidy=2):
    """The reference's own command line on a fixture: stdout and the five output files, kept as golden data.  With missing data
    the run is only kept when every phantom allele slot came out at its floor (printed 0.000000 in *.pklm.txt) and the process
    ended normally: a run in which heap garbage happened to equal an allele code fits a different model (and usually aborts in
    free()).  What was checked is recorded in tests/golden/CLI_MANIFEST.json."""
    out = os.path.join(GOLD, "cli_" + name)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    side = [a for a in args if os.path.isabs(a)]            # -P / -Q files
    phantom = phantom_slots(stru, ploidy)
    env = dict(FIXED_ENV)
    for attempt in range(1, 9):
        tmp, local = scratch_copy([stru] + side, attempt)
        work = os.path.join(tmp, "o")
        os.makedirs(work)
        largs = [local[1 + side.index(a)] if a in side else a for a in args]
        cmd = [REFBIN, "-f", local[0], "-d", work] + largs
        print(" ".join(cmd))
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=work, env=env, text=True)
        bad = None
        if res.returncode != 0:
            bad = "exit status %d" % res.returncode
        else:
            for fn in os.listdir(work):
                if fn.endswith("pklm.txt"):
                    for row in open(os.path.join(work, fn)).read().strip().split("\n")[1:]:
                        k, l, m, v = row.split()
                        if phantom.get(int(l)) == int(m) and v != "0.000000":
                            bad = "phantom slot of locus %s holds %s" % (l, v)
        if bad is None:
            break
        print("   attempt %d rejected: %s" % (attempt, bad))
        shutil.rmtree(tmp)
    else:
        sys.exit("cli_%s: no clean run in 8 attempts" % name)
    for fn in os.listdir(work):
        shutil.copyfile(os.path.join(work, fn), os.path.join(out, fn))
    with open(os.path.join(out, "stdout.txt"), "w") as f:      # elapsed CPU time is not data: masked, so that a re-run gives the same bytes
        import re
        f.write(re.sub(r"\d\d:\d\d:\d\d", "HH:MM:SS", res.stdout.replace(local[0], os.path.basename(stru))))
    shutil.rmtree(tmp)
    with open(os.path.join(out, "ARGS.txt"), "w") as f:      # files are named by their base names (they live in tests/golden/data)
        f.write(" ".join(["-f", os.path.basename(stru)] + [os.path.basename(a) if os.path.isabs(a) else a for a in args]) + "\n")
    CLI_RECORD["cli_" + name] = {"loci_with_a_phantom_allele_slot": len(phantom), "attempts": attempt,
                                 "kept": "exit status 0; every phantom slot printed 0.000000 (no heap garbage matched an allele code)"}


def write_interleaved(src, dst, ploidy):
    """Same data as `src` (ploidy consecutive lines per individual) as one interleaved line per individual, with the
    optional inter-marker distance line the reader must skip (read_file.c:70-82, 100-116)."""
    lines = open(src).read().strip().split("\n")
    hdr, rows = lines[0], [l.split() for l in lines[1:]]
    with open(dst, "w") as f:
        f.write(hdr + "\n")
        f.write("-1 " + " ".join("10" for _ in hdr.split()) + "\n")
        for i in range(0, len(rows), ploidy):
            grp = rows[i:i + ploidy]
            vals = []
            for l in range(len(grp[0]) - 2):
                vals += [g[2 + l] for g in grp]
            f.write("%s_%d %s " % (grp[0][0], i, grp[0][1]) + " ".join(vals) + "\n")


def main():
    if not os.path.exists(HARNESS):
        sys.exit("build oracle/_ref/ref_harness first: make -C oracle ref")
    data = os.path.join(GOLD, "data")
    os.makedirs(data, exist_ok=True)

    c1 = os.path.join(data, "c1_tiny.stru")
    write_c1(c1)
    rnd = random.Random(99)
    multi = os.path.join(data, "multi.stru")
    nal = [rnd.choice([2, 3, 4, 5]) for _ in range(60)]
    codes = [sorted(rnd.sample(range(100, 200), nal[l])) for l in range(60)]   # non-contiguous allele codes
    write_stru(multi, 40, 60, 4, 2, nal, seed=1, allele_codes=codes)
    tetra = os.path.join(data, "tetra.stru")
    write_stru(tetra, 30, 50, 3, 4, [rnd.choice([2, 3, 4]) for _ in range(50)], seed=2)
    miss = os.path.join(data, "missing.stru")
    write_stru(miss, 50, 80, 3, 2, [rnd.choice([2, 2, 3]) for _ in range(80)], seed=3, missing=0.03)

    # loci at which EVERY individual is missing (first of a block of 8, last of one, the file's last locus) beside 3 % scattered
    # missing copies: the reference gives such a locus uniquealleles = 0, i.e. no allele column at all
    allmiss = os.path.join(data, "allmiss.stru")
    write_stru(allmiss, 26, 19, 2, 2, [rnd.choice([2, 3, 4]) for _ in range(19)], seed=12, missing=0.03)
    rows = open(allmiss).read().strip().split("\n")
    with open(allmiss, "w") as f:
        f.write(rows[0] + "\n")
        for r in rows[1:]:
            t = r.split()
            for l in (0, 7, 18):
                t[2 + l] = "-9"
            f.write(" ".join(t) + "\n")
    # monomorphic loci (one allele: p = 1 whatever the cluster), some of them with missing copies (one allele + the phantom slot)
    mono = os.path.join(data, "mono.stru")
    write_stru(mono, 21, 26, 3, 2, [rnd.choice([1, 1, 2, 3]) for _ in range(26)], seed=13, missing=0.04)
    # ploidies the other fixtures do not pin to the reference: haploid, triploid (odd: no pairing of copies), hexaploid (4-bit counts)
    hap = os.path.join(data, "haploid.stru")
    write_stru(hap, 33, 30, 2, 1, [rnd.choice([2, 3, 4]) for _ in range(30)], seed=14, missing=0.03)
    tri = os.path.join(data, "triploid.stru")
    write_stru(tri, 20, 28, 3, 3, [rnd.choice([2, 3, 4, 5]) for _ in range(28)], seed=15, missing=0.03)
    hexa = os.path.join(data, "hexaploid.stru")
    write_stru(hexa, 14, 22, 2, 6, [rnd.choice([2, 3, 4]) for _ in range(22)], seed=16)
    # microsatellite-like loci with more than 32 alleles: the device falls back from the sparse individual pass (LDS tiles sized for
    # <= 32 alleles per locus) to the dense pair of kernels
    many = os.path.join(data, "manyallele.stru")
    write_stru(many, 64, 14, 2, 2, [rnd.choice([3, 12, 35, 44]) for _ in range(14)], seed=17, missing=0.02)
    run("c1_admix_k3", c1, 100, "1,2,3,10,100", 5, ["-a", "-k", "3", "-r", "1234567", "-s", "3"], keep_ilm=False)
    run("manyallele_admix_k2", many, 10, "1,2,3,10", 3, ["-a", "-k", "2", "-r", "4", "-s", "3"])
    run("missing_admix_c_k2", miss, 10, "1,2,3,10", 3, ["-a", "-c", "-k", "2", "-r", "5", "-s", "3"])
    run("mono_admix_k3", mono, 10, "1,2,3,10", 3, ["-a", "-k", "3", "-r", "4", "-s", "3"])
    run("haploid_admix_k2", hap, 10, "1,2,3,10", 3, ["-p", "1", "-a", "-k", "2", "-r", "4", "-s", "3"])
    run("triploid_admix_k3", tri, 10, "1,2,3,10", 3, ["-p", "3", "-a", "-k", "3", "-r", "4", "-s", "3"])
    run("hexaploid_admix_k2", hexa, 10, "1,2,3,10", 3, ["-p", "6", "-a", "-k", "2", "-r", "4", "-s", "3"])
    run("hexaploid_mix_k2", hexa, 5, "1,5", 0, ["-p", "6", "-k", "2", "-r", "4"])
    run("allmiss_admix_k2", allmiss, 10, "1,2,3,10", 3, ["-a", "-k", "2", "-r", "3", "-s", "3"])
    run("allmiss_mix_k2", allmiss, 5, "1,5", 0, ["-k", "2", "-r", "3"])
    run("multi_admix_k4", multi, 30, "1,2,3,10,30", 3, ["-a", "-k", "4", "-r", "7", "-s", "3"])
    run("multi_admix_k4_s1", multi, 3, "1,3", 3, ["-a", "-k", "4", "-r", "7", "-s", "1"])
    run("multi_admix_k4_s2", multi, 3, "1,3", 3, ["-a", "-k", "4", "-r", "7", "-s", "2"])
    run("multi_admix_k3_qn1", multi, 3, "1,3", 3, ["-a", "-k", "3", "-r", "7", "-s", "4"])
    run("multi_admix_k3_qn2", multi, 3, "1,3", 3, ["-a", "-k", "3", "-r", "7", "-s", "5"])
    run("multi_admix_k3_qn3", multi, 3, "1,3", 3, ["-a", "-k", "3", "-r", "7", "-s", "6"])
    # step back-tracking (-g n, accel_em.c:67-82): when the extrapolated point is worse than the EM iterate the step is halved
    # towards -1, up to n times
    run("multi_admix_k4_g3", multi, 3, "1,3", 3, ["-a", "-k", "4", "-r", "7", "-s", "3", "-g", "3"])
    run("multi_admix_k4_s1_g2", multi, 3, "1,3", 3, ["-a", "-k", "4", "-r", "7", "-s", "1", "-g", "2"])
    run("multi_admix_k1", multi, 2, "1,2", 0, ["-a", "-k", "1", "-r", "7"])
    run("tetra_admix_k3", tetra, 20, "1,2,3,20", 3, ["-p", "4", "-a", "-k", "3", "-r", "11", "-s", "3"])
    run("missing_admix_k3", miss, 20, "1,2,3,20", 3, ["-a", "-k", "3", "-r", "5", "-s", "3"])
    run("multi_mix_k3", multi, 10, "1,2,3,10", 3, ["-k", "3", "-r", "5", "-s", "3"])
    run("multi_admix_c_k3", multi, 10, "1,2,3,10", 3, ["-a", "-c", "-k", "3", "-r", "5", "-s", "3"])
    run("missing_mix_k2", miss, 5, "1,5", 0, ["-k", "2", "-r", "5"])
    # tight convergence (-E 1e-10), plain EM: the destination of a fit, not its stopping point on a slow tail, is what is
    # compared.  (With -s 3 at this tolerance the reference itself stops: a rounding-level decrease of the log likelihood
    # between two EM steps makes stop() call exit(0), em_alg.c:112-120.)
    run("c1_admix_k3_tight", c1, 1, "1", 0, ["-a", "-k", "3", "-r", "1234567", "-E", "1e-10"], keep_ilm=False)
    run("multi_admix_k4_tight", multi, 1, "1", 0, ["-a", "-k", "4", "-r", "7", "-E", "1e-10"], keep_ilm=False)

    # (no -s with --projection: an extrapolated point is not projected back then, its log likelihood is NaN and the
    # reference's stop() calls exit(0), em_alg.c:106-110)
    # projection off (--projection): unobserved allele columns keep p = 0 for every k -- the phantom slot of loci with
    # missing data, and alleles a bootstrap replicate lacks (rare.stru has many singleton alleles)
    rare = os.path.join(data, "rare.stru")
    rnd2 = random.Random(123)
    write_stru(rare, 24, 70, 3, 2, [rnd2.choice([3, 4, 5, 6]) for _ in range(70)], seed=5)
    run("missing_admix_k3_noproj", miss, 20, "1,2,3,20", 3, ["-a", "-k", "3", "-r", "5", "--projection"])
    run("multi_admix_k4_noproj", multi, 10, "1,3,10", 3, ["-a", "-k", "4", "-r", "7", "--projection"])
    run("rare_admix_k3_noproj", rare, 10, "1,3,10", 3, ["-a", "-k", "3", "-r", "9", "--projection"])
    run("rare_admix_k3_bs", rare, 3, "1,3", 0, ["-a", "-k", "3", "-r", "9"])
    run("tetra_admix_k3_noproj", tetra, 10, "1,3,10", 3, ["-p", "4", "-a", "-k", "3", "-r", "11", "--projection"])
    # a lower bound far below the default (--bound): t^4 underflows in the shared-reciprocal kernels
    run("multi_admix_k4_tinybound", multi, 10, "1,3,10", 3, ["-a", "-k", "4", "-r", "7", "-s", "3", "--bound", "1e-120"])
    # mixture model with enough loci that exp(max_k v_ik) underflows: logL_mixture's scaling branch
    # (log_likelihood.c:209-224) is executed by the reference here
    mixlong = os.path.join(data, "mixlong.stru")
    write_stru(mixlong, 36, 1800, 3, 2, [2] * 1800, seed=8)
    run("mixlong_mix_k3", mixlong, 6, "1,2,6", 3, ["-k", "3", "-r", "5", "-s", "3"], keep_ilm=False)
    run("mixlong_mix_k3_s1", mixlong, 2, "1,2", 3, ["-k", "3", "-r", "5", "-s", "1"], keep_ilm=False)
    # mixture model that converges slowly enough for several accelerated cycles (few loci: soft assignments)
    mixslow = os.path.join(data, "mixslow.stru")
    rnd4 = random.Random(11)
    write_stru(mixslow, 120, 8, 3, 2, [rnd4.choice([2, 3]) for _ in range(8)], seed=11)
    for sch, tag in ((1, "s1"), (2, "s2"), (3, "s3"), (4, "qn1"), (5, "qn2"), (6, "qn3")):
        run("mixslow_mix_k3_" + tag, mixslow, 3, "1,3", 2, ["-k", "3", "-r", "5", "-s", str(sch)], keep_ilm=False)
    # Rand-EM initialisation (harness section 7; the reference's command line cannot select it)
    wide = os.path.join(data, "wide.stru")
    rnd3 = random.Random(321)
    write_stru(wide, 30, 40, 2, 2, [rnd3.choice([2, 3, 4, 5, 6]) for _ in range(40)], seed=6)
    run("multi_admix_k3_randem", multi, 2, "1,2", 0, ["-a", "-k", "3", "-r", "7"])
    run("wide_admix_k2_randem", wide, 2, "1,2", 0, ["-a", "-k", "2", "-r", "3"])
    run("tetra_admix_k3_randem", tetra, 2, "1,2", 0, ["-p", "4", "-a", "-k", "3", "-r", "11"])
    run("missing_admix_k2_randem", miss, 2, "1,2", 0, ["-a", "-k", "2", "-r", "5"])
    run("multi_admix_c_k3_randem", multi, 2, "1,2", 0, ["-a", "-c", "-k", "3", "-r", "5"])
    run("multi_mix_k3_randem", multi, 2, "1,2", 0, ["-k", "3", "-r", "5"])

    # reader-only fixtures: interleaved layout + "-1" line; remapped missing code
    inter = os.path.join(data, "multi_interleaved.stru")
    write_interleaved(multi, inter, 2)
    run("reader_interleaved", inter, 1, "1", 0, ["-a", "-k", "2", "-r", "1"])
    miss99 = os.path.join(data, "missing99.stru")
    with open(miss) as fi, open(miss99, "w") as fo:
        fo.write(fi.read().replace("-9", "99"))
    run("reader_missing99", miss99, 1, "1", 0, ["-a", "-k", "2", "-r", "1", "--missing", "99"])
    for name in ("reader_interleaved", "reader_missing99"):
        d = os.path.join(GOLD, name)
        for fn in os.listdir(d):
            if fn not in ("manifest.json", "geno.u8", "uniquealleles.i32", "locale.i32", "ARGS.txt"):
                os.remove(os.path.join(d, fn))

    # the reference's own command line: stdout + output files
    run_cli("multi_admix_k4", multi, ["-a", "-k", "4", "-r", "7", "-n", "3"])
    run_cli("allmiss_admix_k2", allmiss, ["-a", "-k", "2", "-r", "3", "-n", "2"])
    run_cli("missing_admix_k3_s3", miss, ["-a", "-k", "3", "-r", "5", "-n", "2", "-s", "3"])
    run_cli("multi_mix_k3", multi, ["-k", "3", "-r", "5", "-n", "2"])
    run_cli("multi_admix_c_k3", multi, ["-a", "-c", "-k", "3", "-r", "5", "-n", "2"])
    # (a K range, -1 <a> -2 <b>, aborts inside the reference itself after the first K: "free(): invalid pointer")
    run_cli("tetra_admix_k3", tetra, ["-p", "4", "-a", "-k", "3", "-r", "11", "-n", "2"], ploidy=4)
    # -i beyond where the warm-up loop stops: on convergence em() returns (em_alg.c:73-74); on the -T cap it still enters
    # its do/while, i.e. one more EM step / accelerated cycle (em_alg.c:61-88)
    run_cli("multi_admix_k4_i1000", multi, ["-a", "-k", "4", "-r", "7", "-n", "2", "-i", "1000"])
    run_cli("multi_admix_k4_i1000_T5", multi, ["-a", "-k", "4", "-r", "7", "-n", "2", "-i", "1000", "-T", "5"])
    run_cli("multi_admix_k4_i1000_T5_s3", multi, ["-a", "-k", "4", "-r", "7", "-n", "2", "-i", "1000", "-T", "5", "-s", "3"])
    run_cli("missing_admix_k3_noproj", miss, ["-a", "-k", "3", "-r", "5", "-n", "2", "--projection"])
    # -P / -Q: initial parameters from files (read_file.c:880-959; biallelic loci): config 1 from a perturbed copy of its own
    # first random initialisation, 17 significant digits
    import struct
    def f64(path):
        raw = open(path, "rb").read()
        return struct.unpack("<%dd" % (len(raw) // 8), raw)
    g = os.path.join(GOLD, "c1_admix_k3")
    q0, p0 = f64(os.path.join(g, "q0.f64")), f64(os.path.join(g, "p0.f64"))
    I, L, K, T = 100, 500, 3, 1000
    qf, pf = os.path.join(data, "c1_Q.txt"), os.path.join(data, "c1_P.txt")
    with open(qf, "w") as f:
        for i in range(I):
            f.write(" ".join("%.17g" % q0[i * K + k] for k in range(K)) + "\n")
    with open(pf, "w") as f:
        for l in range(L):
            f.write(" ".join("%.17g" % (0.25 + 0.5 * p0[k * T + 2 * l]) for k in range(K)) + "\n")
    run_cli("c1_admix_k3_PQ", c1, ["-a", "-k", "3", "-r", "1234567", "-n", "1", "-T", "200", "-P", pf, "-Q", qf])
    run_cli("c1_admix_k3_PQ_s3", c1, ["-a", "-k", "3", "-r", "1234567", "-n", "2", "-s", "3", "-P", pf, "-Q", qf])
    with open(os.path.join(GOLD, "CLI_MANIFEST.json"), "w") as f:
        json.dump(CLI_RECORD, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
