"""The reference-side binding INTEGRATION.md shows is a real file (oracle/glue/ref_glue.c): in the build container it is
compiled with -Werror against the reference's own multiclust.h and linked against libmulticlust_host.so (link check only: no
GPU, nothing runs), so that neither side's structs can drift away from the document."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists only in the build container")
def test_reference_side_glue_compiles_and_links():
    res = subprocess.run(["make", "-B", "-C", os.path.join(ROOT, "oracle"), "glue"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout[-3000:]
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_glue.so"))
    for sym in ("mcamd_flatten_genotypes", "mcamd_options", "mcamd_allocate_model_for_k", "mcamd_initialize_model", "mcamd_em",
                "mcamd_fetch_parameters", "mcamd_fetch_expected_counts", "mcamd_log_likelihood", "mcamd_em_e_step",
                "mcamd_bootstrap_model"):
        assert hasattr(lib, sym), sym


def test_integration_document_quotes_the_glue():
    """every function of the glue is named in INTEGRATION.md"""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src = open(os.path.join(ROOT, "oracle", "glue", "ref_glue.c")).read()
    import re
    for fn in re.findall(r"^\w[\w \*]*\b(mcamd_\w+)\(", src, flags=re.M):
        assert fn in doc, fn
