"""`gt-suffixerator-amd` (C host layer + HIP engine) writes the same files as
the reference's `gt suffixerator` on the reference's own fixtures."""
import hashlib
import os
import subprocess

import pytest

import oracle_util as ou
from genometools_amd import _lib

pytestmark = pytest.mark.gpu
GOLDEN = ou.golden()
CLI = os.path.join(_lib.HERE, "gt-suffixerator-amd")


@pytest.fixture(scope="module")
def cli(gpu):
    subprocess.run(["make", "-C", os.path.join(_lib.HERE, "csrc", "host")], check=True,
                   stdout=subprocess.DEVNULL)
    return CLI


@pytest.mark.parametrize("name", ["Atinsert.fna", "Duplicate.fna", "RandomN.fna",
                                  "TTTN.fna", "Verysmall.fna", "Reads1.fna",
                                  "Copysorttest.fna", "sw100K1.fsa"])
def test_cli_writes_reference_files(cli, name, tmp_path):
    e = GOLDEN[name]
    idx = str(tmp_path / "idx")
    subprocess.run([cli, "-" + e["alphabet"], "-suf", "-lcp", "-bwt", "-tis", "-des",
                    "-ssp", "-db", ou.fixture_path(name), "-indexname", idx], check=True)
    for ext in ("suf", "lcp", "llv", "bwt"):
        with open(idx + "." + ext, "rb") as f:
            raw = f.read()
        assert len(raw) == e["tables"][ext]["bytes"], ext
        assert hashlib.md5(raw).hexdigest() == e["tables"][ext]["md5"], ext
    with open(idx + ".prj") as f:
        assert f.read() == e["prj"]
    # sequence-side files written by default (-des -sds -md5 yes)
    for ext in ("des", "sds", "md5"):
        with open(idx + "." + ext, "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == e["seqfiles"][ext]["md5"], ext


def test_cli_fastq_input_and_switches(cli, tmp_path):
    e = GOLDEN["test10_multiline.fastq"]
    idx = str(tmp_path / "fq")
    subprocess.run([cli, "-dna", "-suf", "-lcp", "-bwt", "-des", "no", "-md5", "no", "-db",
                    ou.fixture_path("test10_multiline.fastq"), "-indexname", idx], check=True)
    for ext in ("suf", "lcp", "llv", "bwt"):
        with open(idx + "." + ext, "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == e["tables"][ext]["md5"], ext
    assert not os.path.exists(idx + ".des") and not os.path.exists(idx + ".md5")
    with open(idx + ".sds", "rb") as f:
        assert hashlib.md5(f.read()).hexdigest() == e["seqfiles"]["sds"]["md5"]


def test_cli_suf_only_and_options_without_effect(cli, tmp_path):
    e = GOLDEN["Atinsert.fna"]
    idx = str(tmp_path / "idx")
    subprocess.run([cli, "-dna", "-suf", "-parts", "3", "-dc", "64", "-v", "-db",
                    ou.fixture_path("Atinsert.fna"), "-indexname", idx], check=True,
                   stdout=subprocess.DEVNULL)
    with open(idx + ".suf", "rb") as f:
        assert hashlib.md5(f.read()).hexdigest() == e["tables"]["suf"]["md5"]
    assert not os.path.exists(idx + ".lcp")
    prj = dict(l.split("=") for l in open(idx + ".prj").read().splitlines())
    assert prj["averagelcp"] == "0.00" and prj["maxbranchdepth"] == "0"
    assert prj["longest"] == "2529"


def test_cli_error_exit_code_and_message(cli, tmp_path):
    p = tmp_path / "bad.fna"
    p.write_text(">a\nACGTX\n")
    r = subprocess.run([cli, "-dna", "-suf", "-db", str(p), "-indexname",
                        str(tmp_path / "i")], capture_output=True, text=True)
    assert r.returncode == 1
    assert r.stderr.startswith("gt suffixerator: error: illegal character 'X': file \"")


VARIANTS = __import__("json").load(open(os.path.join(ou.GOLDEN_DIR, "golden_variants.json")))


@pytest.mark.parametrize("key", sorted(VARIANTS))
def test_cli_readmodes_and_mirror(cli, key, tmp_path):
    name, d, mir = key.split("|")
    e = VARIANTS[key]
    idx = str(tmp_path / "idx")
    subprocess.run([cli, "-dna", "-suf", "-lcp", "-bwt", "-dir", d, "-db",
                    ou.fixture_path(name), "-indexname", idx] +
                   (["-mirrored"] if mir == "1" else []), check=True)
    for ext in ("suf", "lcp", "llv", "bwt"):
        with open(idx + "." + ext, "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == e["tables"][ext]["md5"], ext
    with open(idx + ".prj") as f:
        assert f.read() == e["prj"]
