"""Host C++ mirror (taxator-tk_amd/host): builds, keeps the reference's CLI surface, and refuses to run without a GPU."""
import os
import subprocess

import pytest

import __graft_entry__ as graft

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "taxator-tk_amd", "bin", "taxator")


@pytest.fixture(scope="module")
def taxator_bin():
    if not os.path.exists(BIN):
        graft.build()
    return BIN


def test_help_and_version(taxator_bin):
    out = subprocess.run([taxator_bin, "--help"], capture_output=True, text=True)
    assert out.returncode == 0
    for flag in ("--algorithm", "--seqid-taxid-mapping", "--query-sequences", "--ref-sequences", "--ref-sequences-index",
                 "--processors", "--logfile", "--dataformat"):  # core/taxator.cpp:267-280
        assert flag in out.stdout
    assert subprocess.run([taxator_bin, "-V"], capture_output=True, text=True).stdout.startswith("1.5.0")


def test_missing_mapping_and_bad_algorithm(taxator_bin):
    r = subprocess.run([taxator_bin], capture_output=True, text=True)
    assert r.returncode == 1 and "Specify a taxonomy mapping file" in r.stdout  # taxator.cpp:328-332
    r = subprocess.run([taxator_bin, "-g", "x", "-a", "kraken"], capture_output=True, text=True)
    assert r.returncode == 1 and "classification algorithm can either be: rpa (default), simple-lca, megan-lca" in r.stdout  # :376


def test_no_gpu_fails_loudly(taxator_bin, golden_dir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    d = os.path.join(golden_dir, "survey_edge")
    env = dict(os.environ, TAXATORTK_TAXONOMY_NCBI=os.path.join(d, "tax"))
    r = subprocess.run([taxator_bin, "-g", os.path.join(d, "mapping.tax"), "-q", os.path.join(d, "query_edge.fna"), "-f",
                        os.path.join(d, "ref.fna")], stdin=open(os.path.join(d, "aln_edge.tsv")), capture_output=True, text=True, env=env)
    assert r.returncode == 1
    assert "An unrecoverable error occurred" in r.stderr and "no HIP device" in r.stderr
    assert "taxator-tk" not in r.stdout  # no prediction line was produced by some fallback
    # the alignment-free models run on the device too: same refusal
    r = subprocess.run([taxator_bin, "-a", "simple-lca", "-g", os.path.join(d, "mapping.tax")], stdin=open(os.path.join(d, "aln_edge.tsv")),
                       capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "no HIP device" in r.stderr and "taxator-tk" not in r.stdout
