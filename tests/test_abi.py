"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/gtamd_esa.h and include/gtamd_encode.h declare; host-only entry points work; compute
entry points fail loudly without a device (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from genometools_amd import _lib

HEADERS = [os.path.join(_lib.ROOT, "include", h) for h in ("gtamd_esa.h", "gtamd_encode.h", "gtamd_pck.h")]


def _declared_symbols():
    found = set()
    for header in HEADERS:
        with open(header) as f:
            text = f.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        found.update(re.findall(r"\b(gtamd_[a-z_0-9]+)\s*\(", text))
    return sorted(found)


def test_library_builds_and_loads():
    _lib.build_library()
    assert os.path.exists(_lib.LIB_PATH)
    _lib.load()


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), name
        assert name in _lib.ABI, "ctypes binding missing for " + name
    assert sorted(_lib.ABI) == declared


def test_prefixlength_matches_reference_values():
    lib = _lib.load()
    # values of the reference, SURVEY.md 8a row a5 / BASELINE.md
    for sigma, n, k in [(4, 11817, 4), (4, 16000000, 9), (4, 256000000, 11),
                        (4, 3000000000, 13), (4, 24000000000, 14),
                        (20, 1000000000, 5)]:
        assert lib.gtamd_recommended_prefixlength(sigma, n) == k, (sigma, n)
    # and the prefixlength the reference wrote into .prj for each fixture
    import oracle_util as ou
    for name, e in ou.golden().items():
        prj = dict(l.split("=") for l in e["prj"].splitlines())
        sigma = 20 if e["alphabet"] == "protein" else 4
        assert lib.gtamd_recommended_prefixlength(sigma, int(prj["totallength"])) \
            == int(prj["prefixlength"]), name


def test_prefixlength_matches_oracle_on_a_sweep():
    import oracle_util as ou
    lib, ora = _lib.load(), ou.lib()
    for sigma in (4, 20):
        for e in range(0, 36):
            for n in (1 << e, (1 << e) + 1, 3 * (1 << e) - 1):
                assert lib.gtamd_recommended_prefixlength(sigma, n) == \
                    ora.ora_recommended_prefixlength(sigma, n), (sigma, n)


def test_no_cpu_fallback():
    lib = _lib.load()
    if lib.gtamd_device_count() > 0:
        pytest.skip("a device is present")
    assert not lib.gtamd_esa_create(0, 1000, 4)
    assert b"no HIP device" in lib.gtamd_esa_last_error()
    assert not lib.gtamd_encoder_create(0, 0)
    assert b"no HIP device" in lib.gtamd_esa_last_error()
    enc = np.zeros(10, dtype=np.uint8)
    suf = np.zeros(11, dtype=np.uint64)
    rc = lib.gtamd_esa_build(enc.ctypes.data, 10, 4, 1, suf.ctypes.data, None,
                             None, None, 0, None, None)
    assert rc == -1
    from genometools_amd import esa
    with pytest.raises(esa.EsaError):
        esa.suffixerator_tables(enc)


def test_exceptions_do_not_cross_the_c_abi():
    """every extern "C" entry point runs behind a try/catch barrier
    (GTAMD_ABI_BEGIN / _END, csrc/esa_common.h): a host container that cannot be
    allocated -- e.g. sized from a bogus device-returned count, the pck2 abort of
    round 2, DESIGN.md 9a -- ends in -1 + message, not in std::terminate()"""
    lib = _lib.load()
    assert lib.gtamd_abi_selftest(0) == 0
    assert lib.gtamd_abi_selftest(1 << 16) == 0
    assert lib.gtamd_abi_selftest((1 << 64) - 1) == -1        # std::length_error
    assert b"gtamd_abi_selftest" in lib.gtamd_esa_last_error()
    assert lib.gtamd_abi_selftest(1 << 62) == -1              # std::bad_alloc
    assert b"memory" in lib.gtamd_esa_last_error()
    # the barrier is in every entry point that has a body of its own
    import re
    for f in ("esa_engine.hip", "esa_encode.hip", "esa_pck.hip", "esa_synth.hip"):
        src = open(os.path.join(_lib.HERE, "csrc", f)).read()
        for m in re.finditer(r'^extern "C" (?!void)[^\n;{]*?\b(gtamd_\w+)\([^;{]*\{\n(.*)$', src, re.M):
            assert "GTAMD_ABI_BEGIN" in m.group(2), (f, m.group(1))
