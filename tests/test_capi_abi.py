"""The C-ABI shared library: it loads without a GPU, exports every symbol the header declares,
and its argument checks answer with status codes (never a crash).  No compute calls here."""

import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pyloo_amd.build import build

    build()
    from pyloo_amd import _capi

    return _capi.load_library()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "pyloo_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pla_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from pyloo_amd import _capi

    names = header_symbols()
    assert len(names) >= 12
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/pyloo_amd.h but not exported"
    assert set(names) == set(_capi.SYMBOLS), "ctypes binding and header disagree"
    assert lib.pla_abi_version() == _capi.ABI_VERSION


def test_tail_count_matches_reference_expression(lib):
    from pyloo_amd import _capi
    from pyloo_amd.base import tail_count_for

    for S in (8, 100, 135, 1000, 2000, 4000, 20000, 99991):
        for reff in (0.25, 0.3, 0.7, 1.0, 1.37, 2.0):
            want = int(np.ceil(min(S / 5.0, 3 * (S / reff) ** 0.5)))  # base.py:139-141
            assert _capi.tail_count(S, reff) == want == tail_count_for(S, reff)
    out = C.c_int64(0)
    assert lib.pla_tail_count(4000, -1.0, C.byref(out)) < 0
    assert b"reff" in lib.pla_last_error()


def test_argument_errors_are_status_codes(lib):
    n = C.c_int(-1)
    rc = lib.pla_device_count(C.byref(n))
    assert rc in (0, -5) and n.value >= 0
    h = C.c_void_p()
    if n.value == 0:
        assert lib.pla_engine_create(0, C.byref(h)) == -5  # PLA_ERR_NODEVICE, loud and clean
        assert not h.value
    assert lib.pla_engine_destroy(None) == 0
    # NULL engine -> PLA_ERR_ARG for every entry point
    assert lib.pla_psis_loo(None, None, 0, 1, 10, 10, 1, 0, 2, 1.0, 0.7, 0, None, None, None, None, None) == -1
    assert lib.pla_importance_weights(None, None, 0, 1, 10, 10, 1, 0, 2, 0, None, None, None) == -1
    assert lib.pla_reduce_pointwise(None, None, None, None, 0, 0.7, 0, None, None) == -1
    assert lib.pla_engine_set_timing(None, 1) == -1
    assert lib.pla_fill_synthetic(None, None, 0, 1, 1, 0, 1, 0.1, 0.5, 0.0, 0.0, None) == -1
    assert lib.pla_last_error()


def test_shipped_library_reads_only_the_path_selectors(lib, monkeypatch):
    """the knobs that change timings or results (phase ablation, skipped fit, grids, priorities, the threshold
    check) exist only in -DPLA_EXPERIMENT builds.  The shipped library names what it reads through pla_env_overrides -- the
    five path selectors -- and the kernel sources ask for an environment variable in no other way."""
    from pyloo_amd import _capi

    for k in list(os.environ):
        if k.startswith("PLA_"):
            monkeypatch.delenv(k)
    assert _capi.env_overrides() == ""
    for k in ("PLA_SKIP_FIT", "PLA_DEBUG_SKIP", "PLA_NO_THRESHOLD_CHECK", "PLA_FUSED", "PLA_WAVE_PRIO", "PLA_FIT_GRID", "PLA_NO_TILE"):
        monkeypatch.setenv(k, "1")
    assert _capi.env_overrides() == ""  # not an experiment build: these are not read, and not reported
    monkeypatch.setenv("PLA_PIPE", "0")
    monkeypatch.setenv("PLA_FORCE_PATH", "1")
    assert _capi.env_overrides() == "PLA_PIPE=0 PLA_FORCE_PATH=1"
    assert lib.pla_env_overrides(None, 0) == -1 and lib.pla_engine_stream_stats(None, None) == -1
    # every getenv of the library: the one helper behind env_flag / exp_flag (pla_launch.h) and the selectors of pla_capi.hip
    allowed = {"PLA_PIPE", "PLA_STREAM_PATIENCE_US", "PLA_FORCE_PATH", "PLA_INGEST_TRANSPOSE", "PLA_INGEST_BLOCK_MB"}
    csrc = os.path.join(ROOT, "pyloo_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, f)).read()
        text = re.sub(r"//[^\n]*", "", text)
        for name in re.findall(r'getenv\("(PLA_[A-Z_]+)"\)', text) + re.findall(r'env_flag\("(PLA_[A-Z_]+)"\)', text):
            assert name in allowed, (f, name)
        for name in re.findall(r'exp_(?:flag|str)\("(PLA_[A-Z_]+)"\)', text):
            assert name not in allowed, (f, name)
    text = open(os.path.join(csrc, "pla_launch.h")).read()
    assert "constexpr int exp_flag(const char*) { return 0; }" in text and "#if defined(PLA_EXPERIMENT)" in text


def test_no_cpu_fallback_without_gpu():
    import pyloo_amd
    from pyloo_amd import _capi

    if _capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pyloo_amd.psislw(np.zeros((3, 100)))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pyloo_amd.loo_from_matrix(np.zeros((3, 100)))
