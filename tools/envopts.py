"""Tools only: translate the V3D_* environment switches of the round-1 A/B scripts into the explicit options of the
C-ABI (v3d_sgbm_set_option / v3d_set_option).  The library itself reads no environment variables."""
import os

_SGBM = {"V3D_VDD": "lockstep", "V3D_HFUSED": "hfused", "V3D_HSPLIT": "hsplit", "V3D_CHAIN_DPL": "chain_dpl", "V3D_VDD_DPL": "vdd_dpl",
         "V3D_COST_BAND": "cost_band", "V3D_COST_XCD": "cost_xcd", "V3D_VDD_XCD": "vdd_xcd", "V3D_HF_XCD": "hf_xcd", "V3D_HF_PERSIST": "hf_persist", "V3D_LRM_TILES": "lrm_tiles",
         "V3D_RESERVE_CUS": "reserve_cus", "V3D_VDD_SPIN_LIMIT": "vdd_spin_limit"}
_LIB = {"V3D_GF_BAND1": "gf_band1", "V3D_GF_BAND2": "gf_band2", "V3D_GF_TILED": "gf_tiled", "V3D_GF_FUSED": "gf_fused",
        "V3D_CORR_GATHER": "corr_gather"}


def sgbm_options(env=None):
    """{option: int} for StereoSGBM(options=...) from the V3D_* variables that are set"""
    env = os.environ if env is None else env
    return {opt: int(env[var]) for var, opt in _SGBM.items() if var in env}


def apply_lib_options(native, env=None):
    env = os.environ if env is None else env
    for var, opt in _LIB.items():
        if var in env:
            native.set_option(opt, int(env[var]))


def select_variant_lib(native, env=None):
    """tools only: V3D_HIP_LIB=path selects an experiment build (tools/build_variant.sh) before the library is first loaded"""
    env = os.environ if env is None else env
    if env.get("V3D_HIP_LIB"):
        native.use_library(env["V3D_HIP_LIB"])
