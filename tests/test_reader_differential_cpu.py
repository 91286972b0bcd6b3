"""The STRUCTURE reader against the reference's own read_file(), live, on drawn files: ploidy 1-6, up to 30 alleles per locus with
non-contiguous allele codes, missing values under three different missing codes, interleaved and line-per-copy layouts.
oracle/_ref/ref_harness (ref_harness.c linked with the reference's unmodified sources where /root/reference exists) dumps what
the reference's reader produced -- allele counts per locus, the flat genotype in allele indices, sampling locales -- and the
threaded reader of multiclust_amd/host/mc_reader.c must give the same arrays.  With missing values the reference leaves an
allele slot uninitialised; the harness refuses a run in which the garbage matched an allele code, and such a draw is skipped.
Runs without a GPU; skipped where the harness is absent."""
import json
import os
import random
import subprocess

import numpy as np
import pytest

from test_reader_cpu import read

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
pytestmark = pytest.mark.skipif(not os.access(HARNESS, os.X_OK), reason="oracle/_ref/ref_harness not built")


def write_file(path, rnd, I, L, ploidy, missing_code, missing_rate, interleaved):
    codes = []
    for l in range(L):
        M = rnd.choice([1, 2, 2, 3, 4, 5, 8, 30])
        codes.append(sorted(rnd.sample(range(1, 400), M)))
    npop = rnd.choice([1, 2, 3, 5])
    with open(path, "w") as f:
        f.write(" ".join("L%d" % l for l in range(L)) + "\n")
        for i in range(I):
            rows = []
            for a in range(ploidy):
                rows.append([str(missing_code) if rnd.random() < missing_rate else str(rnd.choice(codes[l])) for l in range(L)])
            head = "id%d p%d " % (i, i % npop)
            if interleaved:
                f.write(head + " ".join(rows[a][l] for l in range(L) for a in range(ploidy)) + "\n")
            else:
                for a in range(ploidy):
                    f.write(head + " ".join(rows[a]) + "\n")


def draw_cases(n, seed):
    rnd = random.Random(seed)
    return [(c, rnd.randrange(2, 40), rnd.randrange(1, 60), rnd.choice([1, 2, 2, 3, 4, 6]), rnd.choice([-9, -9, -1, 0]),
             rnd.choice([0.0, 0.0, 0.02, 0.3]), rnd.choice([0, 0, 1]), rnd.randrange(10 ** 6)) for c in range(n)]


@pytest.mark.parametrize("c,I,L,ploidy,missing_code,missing_rate,interleaved,seed", draw_cases(int(os.environ.get("MC_READER_CASES", "40")), 7))
def test_reader_against_the_reference_reader_on_drawn_files(c, I, L, ploidy, missing_code, missing_rate, interleaved, seed, tmp_path):
    if interleaved and ploidy == 1:
        interleaved = 0                                   # one line per individual either way
    K = 2
    if I < K:
        I = K
    stru = str(tmp_path / "f.stru")
    write_file(stru, random.Random(seed), I, L, ploidy, missing_code, missing_rate, interleaved)
    out = tmp_path / "o"
    out.mkdir()
    cmd = [HARNESS, str(out), "1", "1", "0", "--", "-f", stru, "-a", "-k", str(K), "-p", str(ploidy), "-r", "3", "--missing", str(missing_code)]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    if res.returncode != 0 and ("phantom allele slot has counts" in res.stderr or res.returncode < 0):
        pytest.skip("heap garbage in the reference's uninitialised allele slot: %s" % res.stderr.strip()[-80:])
    rc, d = read(stru, ploidy=ploidy, missing=missing_code)
    if res.returncode != 0:
        assert rc != 0, res.stderr[-300:]                  # a file the reference refuses, this reader refuses
        return
    assert rc == 0
    text = open(str(out / "manifest.json")).read()
    try:
        m = json.loads(text)
    except json.JSONDecodeError:
        # the EM steps the harness goes on to run ended in one of the reference's exit(0)s (a NaN on a file of four individuals):
        # everything the reader produced was written before that
        m = json.loads(text + "\n}")
    assert (d["I"], d["L"], d["ploidy"], d["T"], d["M"]) == (m["I"], m["L"], m["ploidy"], m["T"], m["M"])
    assert d["missing_data"] == m["missing_data"] and d["numpops"] == m["numpops"]
    assert np.array_equal(d["ua"], np.fromfile(str(out / "uniquealleles.i32"), dtype=np.int32))
    assert np.array_equal(d["geno"].ravel(), np.fromfile(str(out / "geno.u8"), dtype=np.uint8))
    assert np.array_equal(d["locale"], np.fromfile(str(out / "locale.i32"), dtype=np.int32))
