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


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists only in the build container")
def test_bound_reference_program_links_and_has_no_cpu_path(tmp_path):
    """oracle/_ref/multiclust_ref_hip (the reference's driver, parser, reader and writers on our EM layer: oracle/glue/ref_bind.c)
    builds; without a GPU it parses its command line, reads the data and then stops in allocate: there is no CPU fallback."""
    res = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "glue"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout[-3000:]
    exe = os.path.join(ROOT, "oracle", "_ref", "multiclust_ref_hip")
    undefined = subprocess.run(["nm", "-u", exe], stdout=subprocess.PIPE, text=True).stdout
    assert "mc_em" in undefined and "mc_model_create" in undefined            # the EM layer comes from libmulticlust_host.so
    defined = subprocess.run(["nm", "--defined-only", exe], stdout=subprocess.PIPE, text=True).stdout
    for sym in ("e_step_admixture_orig", "m_step_admixture_orig", "accelerated_em_step", "logL_admixture", "michelot_project"):
        assert sym not in defined, sym                                          # none of the reference's EM layer is inside
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_gpu_refbind.py runs the program")
    stru = os.path.join(ROOT, "tests", "golden", "data", "multi.stru")
    run = subprocess.run([exe, "-f", stru, "-d", str(tmp_path), "-a", "-k", "4", "-r", "7", "-n", "1"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=120)
    assert "no device context (there is no CPU fallback)" in run.stderr and "initialization = 0" not in run.stdout


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists only in the build container")
def test_libc_rand_stream_on_loan(tmp_path):
    """ref_bind.c borrows glibc's rand() state for an initialisation and hands it back advanced: interleaving rand() with draws
    on the borrowed state gives the draws of one uninterrupted rand() sequence (tests/stream_loan_check.c)."""
    exe = str(tmp_path / "slc")
    lib = os.path.join(ROOT, "multiclust_amd", "lib")
    res = subprocess.run(["gcc", "-std=c17", "-O1", "-w", "-I" + REF, "-I" + os.path.join(ROOT, "multiclust_amd", "host"),
                          "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle", "glue"), "-o", exe,
                          os.path.join(ROOT, "tests", "stream_loan_check.c"), os.path.join(ROOT, "oracle", "glue", "ref_glue.c"),
                          "-L" + lib, "-lmulticlust_host", "-lmulticlust_hip", "-Wl,-rpath," + lib, "-lm"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout[-2000:]
    for seed in ("1", "7", "1234567"):
        out = subprocess.run([exe, seed], stdout=subprocess.PIPE, text=True, timeout=60).stdout.strip()
        assert out == "ok", out

