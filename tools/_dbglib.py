"""Measurement tools that rely on environment overrides of kernel selection (LA_GEMM_PATH, LA_GEMM_NO_PERSISTENT, ...) or on the
ablation bits of la_gemm_variant need the -DLA_DEBUG library: `make -C labelanything_amd/csrc DEBUG=1` -> libla_hip_dbg.so.
Import this module BEFORE labelanything_amd to route the ctypes binding to it."""
import os

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DBG = os.path.join(_HERE, "labelanything_amd", "libla_hip_dbg.so")


def use_debug_library() -> str:
    if not os.path.exists(_DBG):
        raise RuntimeError("build the measurement library first: make -C labelanything_amd/csrc DEBUG=1")
    os.environ["LA_HIP_LIB"] = _DBG
    return _DBG
