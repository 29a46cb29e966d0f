"""The C-ABI library loads on a CPU-only box and exports every symbol include/v3d_hip.h declares.
(No compute calls here: those need a GPU and live in the -m gpu tests.)"""
import ctypes
import os
import re

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "v3d_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(v3d_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from video_3d_pipeline import _native
    lib = ctypes.CDLL(_native.lib_path())
    declared = _declared_symbols()
    assert len(declared) >= 20
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"declared in v3d_hip.h but not exported: {missing}"
    assert sorted(_native.EXPORTS) == declared, "python binding list and header disagree"


def test_no_torch_types_in_the_abi():
    text = open(os.path.join(ROOT, "include", "v3d_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)                 # signatures only, comments stripped
    assert "torch" not in code.lower() and "at::" not in code and "Tensor" not in code and "#include <hip" not in code


def test_version_and_error_strings():
    from video_3d_pipeline import _native
    lib = _native.lib()
    assert b"gfx950" in lib.v3d_version()
    assert isinstance(lib.v3d_last_error(), bytes)
    p = _native.default_params()
    assert (p.numDisparities, p.blockSize, p.P1, p.P2, p.mode) == (64, 5, 600, 2400, 0)
    assert lib.v3d_guided_upscale_ws_bytes(3840, 2160) == 3840 * 2160 * 8 * 2
    assert lib.v3d_corr_ws_bytes(256, 270, 480) == 256 * 270 * 480 * 2
    assert lib.v3d_sgbm_profile_stage_count() == 13


def test_library_reads_no_environment():
    """tuning goes through v3d_sgbm_set_option / v3d_set_option (declared in the header), never getenv in the C-ABI"""
    csrc = os.path.join(ROOT, "video-3d-pipeline_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp", ".h")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f"{f} reads the environment"
    assert {"v3d_sgbm_set_option", "v3d_sgbm_get_option", "v3d_set_option"} <= set(_declared_symbols())
    from video_3d_pipeline import _native
    lib = _native.lib()
    assert lib.v3d_set_option(b"no_such_switch", 1) == -1 and b"unknown option" in lib.v3d_last_error()
    assert lib.v3d_set_option(b"gf_band1", 90) == 0
    assert lib.v3d_sgbm_set_option(None, b"lockstep", 1) == -1            # null handle is an argument error, not a crash
    # every library-wide switch reads back (v3d_get_option), bad values are refused and leave the setting alone
    import ctypes as C
    v = C.c_int(-7)
    for key, good, bad in ((b"gf_band", 128, 3), (b"gf_cols", 512, 300), (b"gf_int1", 0, None), (b"gf_fused", 0, None),
                           (b"corr_fused", 0, None), (b"gf_band1", 60, 1 << 20)):
        assert lib.v3d_get_option(key, C.byref(v)) == 0
        before = v.value
        assert lib.v3d_set_option(key, good) == 0 and lib.v3d_get_option(key, C.byref(v)) == 0 and v.value == good
        if bad is not None:
            assert lib.v3d_set_option(key, bad) == -1 and lib.v3d_get_option(key, C.byref(v)) == 0 and v.value == good
        assert lib.v3d_set_option(key, before) == 0
    assert lib.v3d_get_option(b"no_such_switch", C.byref(v)) == -1


def test_product_does_not_import_the_oracle():
    """the product path must never route through oracle/ (or any CPU fallback)"""
    pkg = os.path.join(ROOT, "video-3d-pipeline_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "liboracle" not in src, f"{f} references the oracle library"


def test_bench_algorithmic_bytes_match_the_survey():
    """bench.py's per-launch shares must sum to SURVEY.md 8(d): 1 291 161 600 B (SGBM) and 190 771 200 B (guided)"""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    sg, gf = mod.alg_bytes_per_frame()
    assert int(sum(sg.values())) == 1291161600 and gf["guided_sweep1+2"] == 190771200
