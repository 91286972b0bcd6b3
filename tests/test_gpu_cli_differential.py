"""-m gpu: the drop-in command line against the UNMODIFIED reference program, live.  oracle/_ref/multiclust_ref (the reference's
own sources compiled by oracle/Makefile in the build container; the binary travels with the tree, the sources do not) and
multiclust_amd/bin/multiclust run the same freshly generated STRUCTURE files with the same arguments; stdout lines (iteration
counts exactly for plain EM) and the five output files must agree as in tests/test_gpu_cli.py.  Where test_gpu_cli.py compares
with committed outputs of twelve command lines, this draws its cases: ploidy 1-6, K 2-9, up to 36 alleles per locus, admixture / -c / mixture, every
acceleration scheme, -g, -n 2 (two initialisations from one rand() stream), -i, -T, --projection, --bound, -E / -e, and 3 % missing
values in a quarter of the cases (a case is skipped when the reference's uninitialised allele slot spoils ITS run).
Skipped where the reference binary is absent."""
import os
import random
import subprocess
import sys

import pytest

import procutil

import test_gpu_cli as cli

sys.path.insert(0, os.path.join(cli.ROOT, "oracle"))
pytestmark = pytest.mark.gpu
REFBIN = os.path.join(cli.ROOT, "oracle", "_ref", "multiclust_ref")


SIZE = int(os.environ.get("MC_DIFF_SIZE", "1"))        # soaks: individuals and loci multiplied by this
SIZE_LIMIT = SIZE ** 3                                   # (a hexaploid 328 x 236 plain-EM run to -E 1e-6 takes the reference minutes)


def run_program(cmd, cwd, timeout=None):
    """One launch, never a second one (tests/procutil.py): a program that does not return within the limit fails the test, with
    what it had printed, where each of its threads stood and -- for ours, which runs under MC_WATCHDOG_S -- the library call it
    was waiting in.  (One launch in about 4 000 of round 3's soaks, a 60-locus mixture fit of 0.3 s, did not return within 300 s
    and left no such record: DESIGN.md section 9.)"""
    return procutil.run_program(cmd, cwd=cwd, timeout=timeout or 120 * SIZE_LIMIT)



def draw_cases(n, seed):
    rnd = random.Random(seed)
    out = []
    for c in range(n):
        ploidy = rnd.choice([1, 2, 2, 2, 3, 4, 6])
        K = rnd.choice([1, 2, 3, 3, 4, 5, 6, 9])                      # K = 1: em() takes one step and a log likelihood (em_alg.c:49-58)
        model = rnd.choice(["-a", "-a", "-a", "-a -c", ""])           # "" = mixture
        scheme = rnd.choice([0, 0, 1, 2, 3, 3, 4, 5, 6]) if model != "" or rnd.random() < 0.5 else 0
        extra = rnd.choice(["", "", "-n 2", "-T 9", "-i 3", "-n 2 -T 30"])
        I, L = rnd.randrange(24, 90) * SIZE, rnd.randrange(20, 120) * SIZE
        more = rnd.choice(["", "", "", "--projection", "--bound 1e-5", "-E 1e-6", "-e 1e-9 -E 0", "-g 2", "-g 3"])
        if more.startswith("-g") and not 1 <= scheme <= 3:
            more = ""                                                 # step back-tracking belongs to SQUAREM (accel_em.c:67-82, multiclust.c:818-819)
        extra = (extra + " " + more).strip()
        if scheme and "-T" not in extra:
            # an accelerated run left to converge takes hundreds of cycles, and its path is sensitive to the last bit of every sum
            # (tests/test_gpu_host_driver.py::test_squarem_path_depends_on_summation_order; three of these cases run to the end
            # stopped 0.1-0.5 log-likelihood units apart from the reference, 2-30 iterations earlier or later): the first 40
            # iterations are compared here, every cycle of whole reference runs in test_gpu_host_driver.py
            extra = (extra + " -T 40").strip()
        out.append((c, I, L, ploidy, K, model, scheme, extra, rnd.randrange(1, 10 ** 6) * 10 + rnd.choice([0, 0, 0, 3])))
    return out


@pytest.mark.skipif(not os.access(REFBIN, os.X_OK), reason="oracle/_ref/multiclust_ref not built (needs /root/reference at build time)")
# MC_DIFF_CASES / MC_DIFF_SEED: more cases, other draws, for occasional soak runs (defaults are what the suite runs)
@pytest.mark.parametrize("c,I,L,ploidy,K,model,scheme,extra,seed",
                         draw_cases(int(os.environ.get("MC_DIFF_CASES", "24")), 20250117 + int(os.environ.get("MC_DIFF_SEED", "0"))))
def test_command_line_against_the_reference_program_on_drawn_cases(c, I, L, ploidy, K, model, scheme, extra, seed, tmp_path):
    from make_fixtures import write_stru, phantom_slots
    seed, missing = seed // 10, (seed % 10) / 100.0        # 0 or 3 % of the allele copies missing
    rnd = random.Random(seed)
    # one case in five has loci with 36 alleles: beyond the 32 rows per locus of the LDS tiles, the dense fallback kernels run
    alleles = [2, 2, 3, 4, 5, 12, 36] if seed % 5 == 0 else [2, 2, 3, 4, 5]
    stru = str(tmp_path / ("d%d.stru" % c))
    write_stru(stru, I, L, max(2, K - 1), ploidy, [rnd.choice(alleles) for _ in range(L)], seed=seed, missing=missing)
    phantom = phantom_slots(stru, ploidy) if missing else {}
    args = ["-p", str(ploidy), "-k", str(K), "-r", str(seed % 9973 + 1)] + model.split() + extra.split()
    if "-n" not in args:
        args += ["-n", "1"]
    if scheme:
        args += ["-s", str(scheme)]
    outs = {}
    for name, exe in (("ref", REFBIN), ("hip", cli.BIN)):
        d = tmp_path / name
        d.mkdir()
        res = run_program([exe, "-f", stru, "-d", os.path.join(str(d), "")] + args, str(d))
        open(str(tmp_path / (name + ".stdout")), "w").write(res.stdout)        # kept for scripts/diag/diffcase.sh
        open(str(tmp_path / (name + ".stderr")), "w").write(" ".join(args) + "\n" + res.stderr)
        if name == "ref" and phantom:
            # Missing values: the reference's reader counts an allele slot for the missing code and never initialises it
            # (read_file.c:527-533 against 581-585).  When the heap garbage in it equals an allele code the reference fits another
            # model, and depending on the length of its path strings it aborts in free(): such a run is no reference for anything
            # (oracle/make_fixtures.py keeps the same rule for the committed goldens)
            if res.returncode != 0:
                pytest.skip("the reference aborted (uninitialised allele slot): %s" % res.stderr.strip()[-80:])
            for fn in os.listdir(str(d)):
                if fn.endswith("pklm.txt"):
                    for row in open(os.path.join(str(d), fn)).read().strip().split("\n")[1:]:
                        k_, l_, m_, v_ = row.split()
                        if phantom.get(int(l_)) == int(m_) and v_ != "0.000000":
                            pytest.skip("heap garbage in the reference's phantom allele slot matched an allele code")
        assert res.returncode == 0, (name, args, res.stderr[-2000:])
        outs[name] = (cli.CLOCK.sub("HH:MM:SS", res.stdout).strip().split("\n"), d)
    (ref_lines, ref_dir), (got_lines, got_dir) = outs["ref"], outs["hip"]
    assert len(ref_lines) == len(got_lines), (args, ref_lines, got_lines)
    # plain EM: same iteration counts, files to 6 decimals.  Accelerated schemes: the extrapolated path amplifies last-bit
    # differences (tests/test_gpu_host_driver.py: test_squarem_path_depends_on_summation_order), so counts and values get the
    # tolerance test_gpu_cli.py gives its accelerated goldens
    exact = scheme == 0
    atol = 2e-6 if exact else 5e-3
    for r, g in zip(ref_lines, got_lines):
        assert cli.NUM.sub("#", r) == cli.NUM.sub("#", g), (args, r, g)
        toks = cli.NUM.findall(r)
        for tok, x, y in zip(toks, [float(t) for t in toks], [float(t) for t in cli.NUM.findall(g)]):
            if "." not in tok and "e" not in tok and abs(x) < 1e6:
                if exact:
                    # iteration counts, exactly -- except at the end of a very long run to a tight -E: a fit of 6 222 (18 199)
                    # iterations whose last steps gain 1e-6 each crossed the threshold one (two) iterations apart
                    assert x == y or (x > 2000 and abs(x - y) <= 3), (args, r, g)
            else:
                assert abs(x - y) <= max(atol * 10, 1e-6 * abs(x)) + (0 if exact else 5e-2), (args, r, g)
    ours = sorted(f for f in os.listdir(got_dir))
    files = sorted(os.listdir(ref_dir))
    if not files:
        # the reference left through its exit(0) on a NaN log likelihood (em_alg.c:106-110) before it wrote anything: so must we
        assert ours == [] and ref_lines == got_lines == [""], (args, ref_lines, got_lines, ours)
        return
    if model == "":
        # mixture model: whether the reference writes <file>_mix_popq.popq depends on an errno it never cleared -- popq_mix()
        # returns silently when errno is non-zero after its malloc (write_file.c:628-636), and an exp() that underflowed anywhere
        # in the E steps leaves ERANGE there -- so the file is missing, or still that of an earlier initialisation, in about one
        # run in twenty.  Ours is always written (its content is pinned by the golden cli_multi_mix_k3, where the reference wrote it)
        popq = [f for f in ours if f.endswith("_mix_popq.popq")]
        assert len(popq) == 1
        ours = [f for f in ours if f != popq[0]]
        files = [f for f in files if f != popq[0]]
        assert len(files) == 4
    elif "--projection" in extra.split() and scheme and len(files) == 3:
        # projection off AND an extrapolated step: entries of P leave [0, 1] (the reference prints p = -0.000206), a cell's
        # t = sum_k q_k p_k can turn negative, and log() of it leaves EDOM in the errno the reference's popq / indivq writers test
        # after their malloc (write_file.c:417,424,505,512): those two files are then missing.  The run itself is compared:
        # where the log likelihood of the extrapolated point is NaN both programs fall back to the EM iterate, and a NaN in an
        # E step ends both at the same iteration (the kernels' log-product must not let two negative factors cancel -- the two
        # copies of a homozygote included: both found here, in the sparse and in the dense kernels)
        assert all(not f.endswith(("popq", "indivq")) for f in files)
        ours = [f for f in ours if not f.endswith(("popq", "indivq"))]
    else:
        assert len(files) == 5
    assert files == ours, (files, ours)
    if "-c" in model.split():
        # shared mixing proportions: the likelihood is a function of sum_k eta_k p_klm alone, so the maximum is a ridge of (eta, P)
        # with one log likelihood, and along it EM neither contracts nor expands a difference: last-bit differences of the sums
        # drift (a plain-EM run to -E 1e-6 ended with eta_0 = 0.2044 against 0.1946, same log likelihood to the printed digit,
        # same iteration counts; an extrapolated step from a nearly stationary point lands anywhere on the ridge).  The lines
        # with the log likelihoods and iteration counts were compared above; the parameters of this model are not comparable
        return
    for fn in files:
        ref_fn, got_fn = os.path.join(ref_dir, fn), os.path.join(got_dir, fn)
        if fn.endswith("out.txt") and not exact:
            # the cluster sizes under "count.K" are an argmax per individual: a fit whose clusters (nearly) coincide -- -c with K = 2
            # ends on etak = 0.504 / 0.496 and every individual at 0.51 / 0.49 -- assigns by the last bits, which an accelerated
            # path does not reproduce.  Compared for plain EM only
            def without_counts(path):
                rows = open(path).read().split("\n")
                at = rows.index("count.K")
                return rows[:at + 1] + rows[at + 2:]
            a, b = str(tmp_path / "ref_out.txt"), str(tmp_path / "got_out.txt")
            open(a, "w").write("\n".join(without_counts(ref_fn)))
            open(b, "w").write("\n".join(without_counts(got_fn)))
            ref_fn, got_fn = a, b
        cli.compare_file(ref_fn, got_fn, atol if not fn.endswith("out.txt") else max(atol * 10, 5e-2 if not exact else 1e-5))


def run_both(tmp_path, args, stru, status=0):
    outs = {}
    for name, exe in (("ref", REFBIN), ("hip", cli.BIN)):
        d = tmp_path / name
        d.mkdir()
        res = run_program([exe, "-f", stru] + args, str(d), timeout=300)
        open(str(tmp_path / (name + ".stderr")), "w").write(res.stderr)
        assert res.returncode == status, (name, args, res.returncode, res.stderr[-2000:])
        outs[name] = (cli.CLOCK.sub("HH:MM:SS", res.stdout).strip().split("\n"), d)
    return outs["ref"], outs["hip"]


@pytest.mark.skipif(not os.access(REFBIN, os.X_OK), reason="oracle/_ref/multiclust_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("args,n_files", [
    ("-a -k 3 -n 2 -r 5 -o myname", 5),          # -o: the five files take the given name (written where the program runs)
    ("-k 2 -r 5 -n 3 -M -d ./", 4),              # -M: the summary line and the maximum log likelihood, nothing else
    ("-a -1 2 -2 3 -n 1 -r 5 -d ./", 10),        # a range of K: five files per K
    ("-a -k 3 -r 5 -n 3 -w n 2", 0),             # -w n 2: the whole estimation twice, run statistics, no files
    ("-a -1 2 -2 4 -r 9 -n 2 -w n 3 -s 3 -T 20", 0),     # ... over a range of K: the K that AIC and BIC pick, averaged
    ("-k 3 -r 5 -n 2 -w n 2 -u l -4700", 0),    # ... with a target log likelihood: "reach target" counts
    # -b: H0 (K - 1) and HA (K) on the observed data, then replicates drawn from the H0 fit -- on the device here, from the same
    # rand() stream as parametric_bootstrap() -- each fitted under both; test statistics and running p-value
    ("-a -k 3 -n 2 -b 3 -r 9 -d ./", 10),
    ("-a -k 4 -n 1 -b 2 -r 4 -s 3 -T 30 -d ./", 10),
    ("-k 3 -n 2 -b 2 -r 4 -d ./", 8),            # mixture model: replicates drawn on the host (popq files: see above)
])
def test_bookkeeping_options_against_the_reference_program(args, n_files, tmp_path):
    """the options that only drive maximize_likelihood()'s and timed_model_estimation()'s bookkeeping (multiclust.c:150-345,
    516-653): same lines, same numbers (plain EM: iteration counts exactly), same files.  Seconds are masked."""
    stru = os.path.join(cli.GOLD, "data", "multi.stru")
    (ref_lines, ref_dir), (got_lines, got_dir) = run_both(tmp_path, args.split(), stru)
    compare_bookkeeping_lines(ref_lines, got_lines, stru, "-w" in args.split())
    ref_files, got_files = (sorted(f for f in os.listdir(x) if not f.endswith("_mix_popq.popq")) for x in (ref_dir, got_dir))
    assert ref_files == got_files and len(ref_files) == n_files, (ref_files, got_files)
    for fn in ref_files:
        cli.compare_file(os.path.join(ref_dir, fn), os.path.join(got_dir, fn), 2e-6 if not fn.endswith("out.txt") else 1e-5)


def compare_bookkeeping_lines(ref_lines, got_lines, stru, timed):
    assert len(ref_lines) == len(got_lines), (ref_lines, got_lines)
    for r, g in zip(ref_lines, got_lines):
        if timed and r.startswith("Average time:"):
            assert g.startswith("Average time:")
            continue
        rt, gt = r.split(), g.split()
        assert len(rt) == len(gt), (r, g)
        if timed and rt[0] == stru:
            # per-repetition line: print_model_state()'s fields, then cumulative and average seconds, then eight more fields
            for x in (-10, -9):
                rt[x] = gt[x] = "seconds"
        for a, b in zip(rt, gt):
            try:
                fa, fb = float(a.strip("(),;[]")), float(b.strip("(),;[]"))
            except ValueError:
                assert a == b, (r, g)
                continue
            assert abs(fa - fb) <= max(2e-5, 1e-9 * abs(fa)) or (fa != fa and fb != fb), (r, g)


@pytest.mark.skipif(not os.access(REFBIN, os.X_OK), reason="oracle/_ref/multiclust_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("partition", ["locales", "random3", "one", "every_other"])
@pytest.mark.parametrize("args", [
    "-a -k 3 -n 3 -r 5 -d ./",                   # the index of the best initialisation's MAP partition, in print_model_state()'s line
    "-k 3 -n 2 -r 5 -d ./",                      # mixture model: partition_mixture (argmax of vik)
    "-a -1 2 -2 3 -n 2 -r 7 -w n 3",             # -w: no files, so the partition is taken for -A alone; RAND statistics of the repetitions
])
def test_partition_file_and_adjusted_rand_index_against_the_reference_program(partition, args, tmp_path):
    """-A <file>: cluster labels 1, 2, ... of the individuals (read_afile, read_file.c:970-999); whenever an initialisation
    improves the maximum its MAP partition is compared with them (adj_rand, multiclust.c:602-612, 1903-1985) and the index takes
    the place of "ND" in the summary line (728-731); -w reports its average, spread and maximum over the repetitions (252-266,
    320-324).  Same lines as the unmodified reference program, on four partitions: the locales of the data file, three random
    classes, one class (0 / 0: "nan" or "-nan", the sign is not compared), two alternating classes."""
    stru = os.path.join(cli.GOLD, "data", "multi.stru")
    rows = [ln.split() for ln in open(stru).read().strip().split("\n")[1:]]
    rows = [r for r in rows if len(r) > 2 and r[0] != "-1"][::2]                    # one line per individual (diploid, two lines each)
    pops = sorted(set(r[1] for r in rows))
    rnd = random.Random(4)
    labels = {"locales": [pops.index(r[1]) + 1 for r in rows], "random3": [rnd.randrange(3) + 1 for _ in rows],
              "one": [1 for _ in rows], "every_other": [1 + x % 2 for x in range(len(rows))]}[partition]
    afile = str(tmp_path / "partition.txt")
    open(afile, "w").write("\n".join(str(x) for x in labels) + "\n")
    (ref_lines, _), (got_lines, _) = run_both(tmp_path, args.split() + ["-A", afile], stru)
    unsigned = [[ln.replace("-nan", "nan") for ln in lines] for lines in (ref_lines, got_lines)]
    compare_bookkeeping_lines(unsigned[0], unsigned[1], stru, "-w" in args.split())
    assert any(" ND " not in ln for ln in got_lines if ln.startswith(stru))         # the index is on the summary line


@pytest.mark.skipif(not os.access(REFBIN, os.X_OK), reason="oracle/_ref/multiclust_ref not built (needs /root/reference at build time)")
def test_bootstrap_whose_null_model_fits_as_well_as_the_alternative_ends_like_the_reference(tmp_path):
    """-c: every K has the same maximum (the likelihood depends on sum_k eta_k p_klm alone), so sooner or later -- here in the first
    replicate -- H0 is not beaten: both programs print what they have fitted until then, say "Null hypothesis likelihood exceeds
    alternative hypothesis likelihood" and leave with INTERNAL_ERROR = 13 (multiclust.c:437-443, message.h:34)."""
    stru = os.path.join(cli.GOLD, "data", "multi.stru")
    (ref_lines, _), (got_lines, _) = run_both(tmp_path, "-a -c -k 3 -n 1 -b 2 -r 4 -d ./".split(), stru, status=13)
    assert len(ref_lines) == len(got_lines) >= 4
    for r, g in zip(ref_lines, got_lines):
        assert cli.NUM.sub("#", r) == cli.NUM.sub("#", g), (r, g)



@pytest.mark.skipif(not os.access(REFBIN, os.X_OK), reason="oracle/_ref/multiclust_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("variant,args", [
    ("interleaved", "-a -k 3 -n 1 -r 5 -d ./"),        # one line per individual, detected from the first two names (read_file.c:84-95)
    ("R", "-a -k 3 -n 1 -r 5 -R -d ./"),               # -R: the header names the two leading columns too (read_file.c:58-59)
    ("plain", "-a -k 3 -n 1 -r 5 -T 6 -v 4 -d ./"),    # -v 4: one line per iteration on stderr (em_alg.c:123-136)
    ("plain", "-a -k 3 -n 1 -r 5 -T 8 -v 4 -s 3 -d ./"),       # ... naming the accelerated steps
    ("plain", "-a -k 3 -n 1 -r 5 -T 8 -v 4 -s 5 -d ./"),       # ... quasi-Newton with two secant pairs: "Q2"
    ("plain", "-k 3 -n 1 -r 5 -T 6 -v 4 -d ./"),               # mixture model
])
def test_input_formats_and_iteration_lines_against_the_reference_program(variant, args, tmp_path):
    src = os.path.join(cli.GOLD, "data", "multi_interleaved.stru" if variant == "interleaved" else "multi.stru")
    stru = src
    if variant == "R":
        rows = open(src).read().split("\n")
        stru = str(tmp_path / "multi_R.stru")
        open(stru, "w").write("\n".join(["id pop " + rows[0]] + rows[1:]))
    (ref_lines, ref_dir), (got_lines, got_dir) = run_both(tmp_path, args.split(), stru)
    assert len(ref_lines) == len(got_lines)
    for r, g in zip(ref_lines, got_lines):
        assert cli.NUM.sub("#", r) == cli.NUM.sub("#", g), (r, g)
        for x, y in zip(cli.NUM.findall(r), cli.NUM.findall(g)):
            assert abs(float(x) - float(y)) <= max(2e-5, 1e-6 * abs(float(x))), (r, g)
    if "-v" in args.split():
        import re
        it = re.compile(r"^\s*\d+ \(")
        ref_it = [l for l in open(str(tmp_path / "ref.stderr")).read().split("\n") if it.match(l)]
        got_it = [l for l in open(str(tmp_path / "hip.stderr")).read().split("\n") if it.match(l)]
        assert len(ref_it) == len(got_it) >= 4, (ref_it, got_it)
        for r, g in zip(ref_it, got_it):
            assert cli.NUM.sub("#", r) == cli.NUM.sub("#", g), (r, g)            # "   3 (S3): # (delta): #"
            for x, y in zip(cli.NUM.findall(r), cli.NUM.findall(g)):
                assert abs(float(x) - float(y)) <= max(0.0051, 2e-5 * abs(float(x))), (r, g)     # two decimals / five significant digits
    files = sorted(os.listdir(ref_dir))
    assert files == sorted(os.listdir(got_dir)) and len(files) in (4, 5)
    for fn in files:
        if not fn.endswith("_mix_popq.popq"):
            cli.compare_file(os.path.join(ref_dir, fn), os.path.join(got_dir, fn), 2e-6 if not fn.endswith("out.txt") else 1e-5)


@pytest.mark.skipif(not os.access(REFBIN, os.X_OK), reason="oracle/_ref/multiclust_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("I,L,ploidy,K,args", [
    (400, 3000, 2, 8, "-a -n 1 -r 12 -T 60"),               # 60 plain EM iterations of a fit with several workgroups per pass
    (300, 2500, 2, 8, "-a -n 1 -r 12 -s 3 -T 30"),          # SQUAREM-3 in the batched device loop
    (250, 1500, 4, 7, "-a -n 2 -r 8 -T 25"),                # tetraploid, two initialisations
])
def test_larger_fits_against_the_reference_program(I, L, ploidy, K, args, tmp_path):
    """the same comparison at sizes beyond the drawn cases (the reference's reader sorts all haplotypes per locus: a few seconds here)"""
    from make_fixtures import write_stru
    rnd = random.Random(I * L)
    stru = str(tmp_path / "big.stru")
    write_stru(stru, I, L, K - 1, ploidy, [rnd.choice([2, 3, 4]) for _ in range(L)], seed=I + L)
    (ref_lines, ref_dir), (got_lines, got_dir) = run_both(tmp_path, ["-p", str(ploidy), "-k", str(K), "-d", "./"] + args.split(), stru)
    assert len(ref_lines) == len(got_lines)
    exact = "-s" not in args.split()
    for r, g in zip(ref_lines, got_lines):
        assert cli.NUM.sub("#", r) == cli.NUM.sub("#", g), (r, g)
        toks = cli.NUM.findall(r)
        for tok, x, y in zip(toks, [float(t) for t in toks], [float(t) for t in cli.NUM.findall(g)]):
            if "." not in tok and "e" not in tok and abs(x) < 1e6:
                assert x == y, (r, g)
            else:
                assert abs(x - y) <= max(2e-5, 1e-6 * abs(x)) + (0 if exact else 5e-2), (r, g)
    files = sorted(os.listdir(ref_dir))
    assert files == sorted(os.listdir(got_dir)) and len(files) == 5
    for fn in files:
        cli.compare_file(os.path.join(ref_dir, fn), os.path.join(got_dir, fn), (2e-6 if exact else 5e-3) if not fn.endswith("out.txt") else (1e-5 if exact else 5e-2))


@pytest.mark.skipif(not os.access(REFBIN, os.X_OK), reason="oracle/_ref/multiclust_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("c,I,L,K,ploidy,scheme,model", [
    (0, 60, 300, 3, 2, 3, "-a"), (1, 60, 300, 3, 2, 1, "-a"), (2, 60, 300, 3, 2, 4, "-a"), (3, 80, 200, 4, 2, 3, "-a"),
    (4, 50, 400, 2, 4, 3, "-a"), (5, 60, 300, 3, 2, 3, ""), (6, 60, 300, 3, 2, 2, "-a"), (7, 90, 250, 3, 2, 5, "-a"),
    (8, 90, 250, 3, 2, 6, "-a"),
])
def test_converged_accelerated_fits_end_at_the_reference_programs_maximum(c, I, L, K, ploidy, scheme, model, tmp_path):
    """Accelerated fits LEFT TO CONVERGE (-E 1e-8, 500-3 500 iterations), every scheme: the two programs take different numbers of
    iterations (578 against 618, 1 923 against 1 736: the extrapolation amplifies the last bit of every sum, which is why the drawn
    cases above compare the first 40 iterations at 5e-3), but both contract to the same maximum, so the END of the run is
    comparable far more tightly than its path: log likelihood within 1e-4 of 2e4-6e4 (seen: 1e-5), every entry of the five files
    within 5e-4 (seen: 1.7e-4 in Q, 9e-5 in P).  Data with a clear structure (each individual 0.8 / 0.2 around its source
    population): a single maximum near the truth.  Tighter criteria are not comparable at all: at -E 1e-10 the reference leaves
    through its "log likelihood decrease" exit (em_alg.c:116) on a decrease of 4e-15 in two of nine such runs and writes nothing."""
    from make_fixtures import write_stru
    rnd = random.Random(c)
    stru = str(tmp_path / "conv.stru")
    write_stru(stru, I, L, K, ploidy, [rnd.choice([2, 2, 3, 4]) for _ in range(L)], seed=100 + c)
    args = ["-p", str(ploidy), "-k", str(K), "-r", "77", "-n", "1", "-s", str(scheme), "-T", "5000", "-E", "1e-8", "-d", "./"] + model.split()
    (ref_lines, ref_dir), (got_lines, got_dir) = run_both(tmp_path, args, stru)
    assert len(ref_lines) == len(got_lines)
    for r, g in zip(ref_lines, got_lines):
        assert cli.NUM.sub("#", r) == cli.NUM.sub("#", g), (r, g)
        assert "(converged)" in r or "initialization" not in r, r
        toks = cli.NUM.findall(r)
        for tok, x, y in zip(toks, [float(t) for t in toks], [float(t) for t in cli.NUM.findall(g)]):
            if "." in tok or "e" in tok:
                assert abs(x - y) <= 1e-4, (r, g)           # log likelihoods, AIC, BIC
    # (mixture model: whether the reference writes its popq file depends on a stale errno, see the drawn cases above)
    files = sorted(f for f in os.listdir(ref_dir) if not f.endswith("_mix_popq.popq"))
    assert files == sorted(f for f in os.listdir(got_dir) if not f.endswith("_mix_popq.popq")) and len(files) in (4, 5)
    for fn in files:
        cli.compare_file(os.path.join(ref_dir, fn), os.path.join(got_dir, fn), 5e-4)


@pytest.mark.skipif(not os.access(REFBIN, os.X_OK), reason="oracle/_ref/multiclust_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("c", range(8))
def test_starting_values_from_files_against_the_reference_program(c, tmp_path):
    """-P / -Q: the admixture fit starts from parameters read from two text files (biallelic data: one frequency per locus and
    cluster, read_file.c:880-959) instead of a random partition; drawn data, drawn parameters, plain EM and accelerated schemes,
    individual and shared (-c) mixing proportions."""
    from make_fixtures import write_stru
    rnd = random.Random(900 + c)
    I, L, K = rnd.randrange(20, 80), rnd.randrange(20, 150), rnd.choice([2, 3, 4, 5])
    scheme, constrained = rnd.choice([0, 0, 3, 1, 5]), c % 4 == 3
    stru = str(tmp_path / "b.stru")
    write_stru(stru, I, L, K, 2, [2] * L, seed=77 + c)
    with open(str(tmp_path / "Q.txt"), "w") as f:
        for _ in range(1 if constrained else I):
            g = [rnd.gammavariate(1.0, 1.0) + 0.05 for _ in range(K)]
            f.write(" ".join("%.17g" % (x / sum(g)) for x in g) + "\n")
    with open(str(tmp_path / "P.txt"), "w") as f:
        for _ in range(L):
            f.write(" ".join("%.17g" % rnd.uniform(0.05, 0.95) for _ in range(K)) + "\n")
    args = ["-a"] + (["-c"] if constrained else []) + ["-k", str(K), "-n", "1", "-r", "3", "-T", "40", "-d", "./",
            "-P", str(tmp_path / "P.txt"), "-Q", str(tmp_path / "Q.txt")] + (["-s", str(scheme)] if scheme else [])
    rows = [l.split()[2:] for l in open(stru).read().strip().split("\n")[1:]]
    if any(len(set(r[l] for r in rows)) < 2 for l in range(L)):
        # a locus at which the sample shows one allele only: the reference stores 1 - p in a second allele slot that does not
        # exist (read_file.c:950, past the allocation) and carries on; this build refuses the combination
        d = tmp_path / "hip"
        d.mkdir()
        res = run_program([cli.BIN, "-f", stru] + args, str(d), timeout=120)
        assert res.returncode == 11 and "-P needs two alleles at every locus" in res.stderr
        return
    (ref_lines, ref_dir), (got_lines, got_dir) = run_both(tmp_path, args, stru)
    assert len(ref_lines) == len(got_lines)
    exact = scheme == 0
    for r, g in zip(ref_lines, got_lines):
        assert cli.NUM.sub("#", r) == cli.NUM.sub("#", g), (r, g)
        toks = cli.NUM.findall(r)
        for tok, x, y in zip(toks, [float(t) for t in toks], [float(t) for t in cli.NUM.findall(g)]):
            if "." not in tok and "e" not in tok and abs(x) < 1e6:
                if exact:
                    assert x == y, (r, g)
            else:
                assert abs(x - y) <= max(2e-5, 1e-6 * abs(x)) + (0 if exact else 5e-2), (r, g)
    files = sorted(os.listdir(ref_dir))
    assert files == sorted(os.listdir(got_dir)) and len(files) == 5
    if constrained:
        return                                               # parameters on the -c ridge are not comparable (see above)
    for fn in files:
        cli.compare_file(os.path.join(ref_dir, fn), os.path.join(got_dir, fn), (2e-6 if exact else 5e-3) if not fn.endswith("out.txt") else (1e-5 if exact else 5e-2))


@pytest.mark.skipif(not os.access(REFBIN, os.X_OK), reason="oracle/_ref/multiclust_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("args", ["-a -k 3 -n 2 -r 6 -T 40", "-a -k 3 -n 1 -r 6 -T 40 -s 3", "-a -k 2 -n 1 -r 6 -T 30 -s 5", "-k 3 -n 1 -r 6"])
def test_a_sample_without_a_single_genotype_against_the_reference_program(args, tmp_path):
    """one individual of the file has the missing code at every locus: the reference prints -nan for its mixing proportions (and
    the fit of everybody else is what it would be without it); accelerated schemes fall back to their EM iterates in every cycle.
    Same lines, same files, "-nan" included.  The reference's uninitialised allele slot (every locus has a missing value here)
    makes some of its runs useless: those are skipped as in the drawn cases."""
    from make_fixtures import write_stru, phantom_slots
    stru = str(tmp_path / "e.stru")
    write_stru(stru, 40, 30, 3, 2, [random.Random(3).choice([2, 3, 4]) for _ in range(30)], seed=12)
    rows = open(stru).read().split("\n")
    for x in (2 * 7 + 1, 2 * 7 + 2):                               # both lines of individual 7
        head = rows[x].split()[:2]
        rows[x] = " ".join(head + ["-9"] * 30)
    open(stru, "w").write("\n".join(rows))
    phantom = phantom_slots(stru, 2)
    outs = {}
    for name, exe in (("ref", REFBIN), ("hip", cli.BIN)):
        d = tmp_path / name
        d.mkdir()
        res = run_program([exe, "-f", stru, "-d", "./"] + args.split(), str(d))
        if name == "ref":
            if res.returncode != 0:
                pytest.skip("the reference aborted (uninitialised allele slot)")
            for fn in os.listdir(str(d)):
                if fn.endswith("pklm.txt"):
                    for row in open(os.path.join(str(d), fn)).read().strip().split("\n")[1:]:
                        k_, l_, m_, v_ = row.split()
                        if phantom.get(int(l_)) == int(m_) and v_ != "0.000000":
                            pytest.skip("heap garbage in the reference's phantom allele slot matched an allele code")
        assert res.returncode == 0, (name, res.stderr[-1000:])
        outs[name] = (cli.CLOCK.sub("HH:MM:SS", res.stdout).strip().split("\n"), d)
    (ref_lines, ref_dir), (got_lines, got_dir) = outs["ref"], outs["hip"]
    assert len(ref_lines) == len(got_lines)
    for r, g in zip(ref_lines, got_lines):
        assert cli.NUM.sub("#", r) == cli.NUM.sub("#", g), (r, g)
        for x, y in zip(cli.NUM.findall(r), cli.NUM.findall(g)):
            assert abs(float(x) - float(y)) <= max(2e-5, 1e-6 * abs(float(x))), (r, g)       # every cycle is two EM steps: exact counts
    files = sorted(f for f in os.listdir(ref_dir) if not f.endswith("_mix_popq.popq"))
    assert files == sorted(f for f in os.listdir(got_dir) if not f.endswith("_mix_popq.popq"))
    n_nan = 0
    for fn in files:
        a, b = open(os.path.join(ref_dir, fn)).read().split(), open(os.path.join(got_dir, fn)).read().split()
        assert len(a) == len(b), fn
        for x, y in zip(a, b):
            if "nan" in x or "nan" in y:
                assert x == y, (fn, x, y)                            # "-nan" where the reference prints "-nan"
                n_nan += 1
                continue
            try:
                assert abs(float(x) - float(y)) <= (2e-6 if not fn.endswith("out.txt") else 1e-5), (fn, x, y)
            except ValueError:
                assert x == y, (fn, x, y)
    if "-a" in args.split():
        assert n_nan >= 1

