"""CPU tests of the STRUCTURE reader (multiclust_amd/host/mc_reader.c) against what the reference's own reader
produced for the same files (golden geno / uniquealleles / locale dumped by oracle/ref_harness.c)."""
import ctypes as C
import os

import numpy as np
import pytest

from golden_util import GOLD, Golden
from multiclust_amd import host


read = host.read_structure          # the binding lives with the other views of the host library


CASES = [("c1_admix_k3", "c1_tiny.stru", 2, -9), ("multi_admix_k4", "multi.stru", 2, -9),
         ("tetra_admix_k3", "tetra.stru", 4, -9), ("missing_admix_k3", "missing.stru", 2, -9),
         ("reader_interleaved", "multi_interleaved.stru", 2, -9), ("reader_missing99", "missing99.stru", 2, 99),
         ("allmiss_admix_k2", "allmiss.stru", 2, -9),
         ("mono_admix_k3", "mono.stru", 2, -9), ("haploid_admix_k2", "haploid.stru", 1, -9), ("triploid_admix_k3", "triploid.stru", 3, -9),
         ("hexaploid_admix_k2", "hexaploid.stru", 6, -9), ("manyallele_admix_k2", "manyallele.stru", 2, -9)]       # three loci at which every individual is missing: no allele column at all


@pytest.mark.parametrize("gold,fn,ploidy,missing", CASES)
def test_reader_matches_reference_reader(gold, fn, ploidy, missing):
    g = Golden(gold)
    rc, d = read(os.path.join(GOLD, "data", fn), ploidy=ploidy, missing=missing)
    assert rc == 0
    assert (d["I"], d["L"], d["ploidy"], d["T"], d["M"]) == (g.I, g.L, g.ploidy, g.T, g.m["M"])
    assert d["missing_data"] == g.m["missing_data"] and d["numpops"] == g.m["numpops"]
    assert np.array_equal(d["ua"], g.ua)
    assert np.array_equal(d["geno"], g.geno)
    assert np.array_equal(d["locale"], g.i32("locale.i32"))
    assert sum(d["i_p"]) == g.I


def test_interleaved_detection_and_names():
    rc, a = read(os.path.join(GOLD, "data", "multi.stru"))
    rc2, b = read(os.path.join(GOLD, "data", "multi_interleaved.stru"))
    assert rc == 0 and rc2 == 0
    assert a["interleaved"] == 0 and b["interleaved"] == 1
    # with the "-1" line the reference's line count is one short (read_file.c:119): the last individual is dropped
    assert a["I"] == 40 and b["I"] == 39
    if np.array_equal(a["ua"], b["ua"]):
        assert np.array_equal(a["geno"][:39], b["geno"])
    assert a["names"][0] == "ind0" and a["pops"][:2] == ["pop0", "pop1"]


def test_r_format_and_errors(tmp_path):
    src = open(os.path.join(GOLD, "data", "multi.stru")).read().split("\n")
    p = tmp_path / "r.stru"
    p.write_text("name pop " + src[0] + "\n" + "\n".join(src[1:]))      # -R: header also names the two info columns
    rc, d = read(str(p), r_format=1)
    assert rc == 0 and d["L"] == 60
    rc, _ = read(str(p), r_format=0)                                        # without -R the column counts disagree
    assert rc != 0
    rc, _ = read(str(tmp_path / "absent.stru"))
    assert rc != 0
    q = tmp_path / "odd.stru"
    q.write_text("\n".join(src[:4]) + "\n")                                 # 3 haplotype lines: not a multiple of ploidy
    rc, _ = read(str(q))
    assert rc != 0
