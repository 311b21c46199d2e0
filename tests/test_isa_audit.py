"""Static audit of the compiled gfx950 ISA for a code-generation fault met in
round 1 (hipcc of ROCm 7.2): after a wave-uniform 64-bit compare the select of
`min(n - tile_base, TILE)` was emitted as

    v_cmp_lt_u64 vcc, ...; s_addc_u32 ...; s_cbranch_vccz ...; s_cselect_b32 valid, rem, 0x1000

i.e. the s_cselect read an SCC that an address addition in between had
overwritten, and the last, partial tile of a sort pass ran as a full one
("Tried and dropped at the end of round 1").  The parity tests catch
the effect on the GPU box; this test catches the pattern here, where hipcc
cross-compiles: every value-selecting s_cselect_b32 must take its SCC from a
compare or a logical operation, not from an addition, subtraction or shift.
(`s_cselect_b64 x, -1, 0` right behind an add/sub materialises a carry -- the
compiler's 64-bit arithmetic -- and is left alone.)"""
import os
import re
import shutil
import subprocess

import pytest

from genometools_amd import _lib

SCC_WRITERS = ("s_cmp", "s_and_", "s_or_", "s_xor_", "s_andn2", "s_orn2", "s_add_", "s_addc",
               "s_sub_", "s_subb", "s_lshl", "s_lshr", "s_ashr", "s_bfe", "s_min", "s_max",
               "s_abs", "s_not", "s_bitcmp", "s_and_saveexec", "s_or_saveexec",
               "s_andn2_saveexec", "s_xor_saveexec", "s_mul_hi")
GOOD = ("s_cmp", "s_and_b", "s_or_b", "s_xor_b", "s_andn2_b", "s_bitcmp", "s_and_saveexec",
        "s_or_saveexec", "s_andn2_saveexec")


def _audit(asm):
    bad, kernel, last = [], "", None
    for line in asm.splitlines():
        m = re.match(r"^(\w+):", line)
        if m and not line.startswith(".L"):
            kernel, last = m.group(1), None
            continue
        if line.startswith(".LBB"):
            last = "label"          # SCC does not survive a join we cannot see through
            continue
        m = re.match(r"^\s+(s_[a-z0-9_]+)\s+(.*)", line)
        if not m:
            continue
        op, args = m.group(1), m.group(2)
        if op == "s_cselect_b32":
            if last is not None and last != "label" and not last.startswith(GOOD):
                bad.append("%s: s_cselect_b32 %s  (SCC from %s)" % (kernel, args.strip(), last))
            continue
        if op.startswith("s_cselect"):
            continue
        if op.startswith(SCC_WRITERS):
            last = op
    return bad


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="needs hipcc")
@pytest.mark.parametrize("src", ["esa_prims.hip", "esa_engine.hip", "esa_encode.hip", "esa_synth.hip"])
def test_value_selects_take_scc_from_a_compare(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = os.path.join(_lib.HERE, "csrc")
    out = str(tmp_path / "k.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S",
                    "--cuda-device-only", "-I", csrc, "-o", out, os.path.join(csrc, src)],
                   check=True, stderr=subprocess.DEVNULL)
    with open(out) as f:
        bad = _audit(f.read())
    assert not bad, "\n".join(bad)


def test_audit_recognises_the_faulty_sequence():
    asm = """k_bad:
\tv_cmp_lt_u64_e32 vcc, s[0:1], v[2:3]
\ts_addc_u32 s25, s5, s3
\ts_cbranch_vccz .LBB12_109
\ts_cselect_b32 s36, s0, 0x1000
k_good:
\tv_cmp_lt_u64_e32 vcc, s[28:29], v[2:3]
\ts_and_b64 s[4:5], vcc, exec
\ts_cselect_b32 s33, s28, 0x1000
\ts_add_u32 s3, s25, s3
\ts_cselect_b64 s[8:9], -1, 0
"""
    bad = _audit(asm)
    assert len(bad) == 1 and bad[0].startswith("k_bad")
