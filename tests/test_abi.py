"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/tsqr_mi.h declares,
and its size helpers / error codes follow the reference (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported(bq):
    hdr = open(os.path.join(ROOT, "include", "tsqr_mi.h")).read()
    declared = set(re.findall(r"\b(tsqr_mi_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = ctypes.CDLL(bq.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(L, sym), "libtsqr_mi.so does not export %s" % sym
    assert declared == set(bq.C_ABI_SYMBOLS)
    assert L.tsqr_mi_version() >= 100


def test_enum_matches_reference(bq):
    # same names, same order as reference src/blockqr.hpp:12-23
    names = ["fp16_notc", "fp16_tc_nocor", "fp32_notc", "fp32_tc_cor", "fp32_tc_nocor", "mixed_tc_cor_emu",
             "tf32_tc_cor", "tf32_tc_cor_emu", "tf32_tc_nocor", "tf32_tc_nocor_emu"]
    assert [m.name for m in bq.compute_mode] == names
    assert [int(m) for m in bq.compute_mode] == list(range(10))
    assert bq.tsqr_colmun_size == 16 and bq.success_factorization == 0 and bq.error_invalid_matrix_size == 1
    hdr = open(os.path.join(ROOT, "include", "tsqr_mi.h")).read()
    for i, n in enumerate(names):
        assert re.search(r"TSQR_MI_%s\s*=\s*%d\b" % (n.upper(), i), hdr)
    hpp = open(os.path.join(ROOT, "include", "tsqr", "blockqr.hpp")).read()
    body = hpp[hpp.index("enum compute_mode"):]
    body = body[:body.index("}")]
    assert re.findall(r"\b([a-z0-9_]+),", body) == names


def test_size_helpers(bq, oracle):
    # batch rule identical to the reference (src/tsqr.cu:39-44); working sizes never below the reference's
    for m in (1, 32, 33, 128, 1999, 9211, 1 << 15, 1 << 20, (1 << 20) + 1, 1 << 23):
        assert bq.get_batch_size_log2(m) == oracle.lib().ref_get_batch_size_log2(m)
        assert bq.get_working_l_size(m) == max(oracle.lib().ref_working_l_size(m), 8)   # never smaller than the reference; >= the 3 status words
        for n in (1, 7, 16, 51, 64, 100, 128):
            if n > m:
                continue
            assert bq.get_working_q_size(m, n) >= oracle.lib().ref_working_q_size(m, n)
            assert bq.get_working_r_size(m, n) >= oracle.lib().ref_working_r_size(m, n)
    # C2 (2^20 x 64): the engine fits in the reference's own workspace sizes (SURVEY.md 8b "Ownership")
    m = 1 << 20
    assert bq.get_working_q_size(m, 64) == oracle.lib().ref_working_q_size(m, 64)
    assert bq.get_working_r_size(m, 64) == oracle.lib().ref_working_r_size(m, 64)
    assert bq.lib().tsqr_mi_working_reorth_size(m) == 512 + 16 * m


def test_invalid_sizes_and_modes_without_gpu(bq):
    # argument checks come before any GPU work (reference src/blockqr.cu:409-411): safe to call without a GPU
    L = bq.lib()
    z = ctypes.c_void_p(0)
    for (m, n) in [(4, 8), (0, 0), (0, 4), (4, 0)]:
        assert L.tsqr_mi_qr_f32(int(bq.compute_mode.fp32_tc_cor), 0, z, 1, z, 1, z, 1, m, n, z, z, z, z, z, z) == 1
    for mode in (bq.compute_mode.fp16_notc, bq.compute_mode.tf32_tc_cor, bq.compute_mode.mixed_tc_cor_emu):
        assert L.tsqr_mi_qr_f32(int(mode), 0, z, 8, z, 4, z, 8, 8, 4, z, z, z, z, z, z) == bq.error_unsupported_mode
    assert "not implemented" in bq.last_error()


def test_missing_library_fails_loudly(monkeypatch):
    from tsqr_gpu_amd import blockqr
    monkeypatch.setattr(blockqr, "_lib", None)
    monkeypatch.setattr(blockqr, "LIB_PATH", "/nonexistent/libtsqr_mi.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        blockqr.lib()
    monkeypatch.undo()


def test_product_does_not_import_oracle():
    # the oracle is test infrastructure: nothing under tsqr_gpu_amd/ may reference it
    pkg = os.path.join(ROOT, "tsqr_gpu_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert "ref_oracle" not in txt and "libref_oracle" not in txt, fn


def test_cpp_header_compiles_and_links(bq):
    """include/tsqr/blockqr.hpp (the mtk::qr surface) compiles with hipcc and links against libtsqr_mi.so."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-s"])
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "sample_blockqr"))
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "sample_tsqr16"))       # include/tsqr/tsqr.hpp: mtk::tsqr::tsqr16 / buffer
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "sample_batch"))        # mtk::qr::qr_batch, get_tsqr_compute_mode<> (static_asserts)
