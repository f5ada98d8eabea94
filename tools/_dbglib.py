"""Measurement tools that rely on environment overrides of kernel selection (LA_GEMM_PATH, LA_GEMM_NO_PERSISTENT, ...) or on the
ablation bits of la_gemm_variant need the -DLA_DEBUG library: `make -C labelanything_amd/csrc DEBUG=1` -> libla_hip_dbg.so.
Call one of these BEFORE the first library call: they re-point ``labelanything_amd._lib.LIB_PATH`` (the product loader itself reads no
environment variable).  ``LA_TOOLS_LIB=<path>`` is the tools-only switch of the same-box A/B scripts (tools/lib_ab.sh, attn_ab.sh)."""
import os
import sys

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DBG = os.path.join(_HERE, "labelanything_amd", "libla_hip_dbg.so")
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)


def _point_at(path: str) -> str:
    from labelanything_amd import _lib
    if _lib._lib is not None and _lib.LIB_PATH != path:
        raise RuntimeError("the HIP library is already loaded: choose the measurement library before the first call")
    _lib.LIB_PATH = path
    return path


def use_debug_library() -> str:
    if not os.path.exists(_DBG):
        raise RuntimeError("build the measurement library first: make -C labelanything_amd/csrc DEBUG=1")
    return _point_at(_DBG)


def use_env_library() -> str:
    """LA_TOOLS_LIB=<path to a libla_hip build>: A/B of two builds from the tools' shell scripts."""
    path = os.environ.get("LA_TOOLS_LIB")
    if not path:
        from labelanything_amd import _lib
        return _lib.LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(f"LA_TOOLS_LIB={path} does not exist")
    return _point_at(os.path.abspath(path))
