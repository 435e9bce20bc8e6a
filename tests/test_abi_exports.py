"""CPU-side checks of the C-ABI library: it loads without a GPU and exports every symbol include/mlqem_hip.h
declares (no compute call is made here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mlqem_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mlqem_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    from blackwater.native import _lib

    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 14
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in include/mlqem_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes signature table and header disagree"
    assert lib.mlqem_abi_version() == _lib.ABI_VERSION
    header = open(os.path.join(ROOT, "include", "mlqem_hip.h")).read()
    assert f"#define MLQEM_ABI_VERSION {_lib.ABI_VERSION} " in header
    assert lib.mlqem_error_string(-4).decode().startswith("workspace")


def test_host_only_entry_points_validate_arguments():
    """Argument validation happens before any launch, so it can be exercised without a device."""
    from blackwater.native import _lib

    lib = _lib.load()
    assert lib.mlqem_linear_f32(None, 4, None, 0, None, None, None, 4, -1, 4, 4, 0, 0, 0.0, 0, -1, -1, None, 0, 1.0, None, None) == -1  # N < 0
    assert lib.mlqem_csr_aggregate_f32(None, 4, None, None, None, None, None, None, 1.0, 0.0, None, 0, None, 0, 1.5, 0,
                                       None, None, 4, 10, 4, None) == -1  # drop_p out of range
    assert lib.mlqem_segment_pool_f32(None, 1, None, None, 0, 0, 1, None, 1, None, 1, None, 0, None) == -1  # no output wanted
    assert lib.mlqem_segment_pool_workspace_bytes(2048, 3, 10) == (2 + 3) * 2 * 12 * 4


def test_ops_refuse_cpu_tensors_loudly():
    import pytest
    import torch

    from blackwater.native import _lib, ops

    x = torch.zeros(4, 3)
    with pytest.raises(_lib.NativeLibraryError, match="no CPU path"):
        ops.linear(x, torch.zeros(2, 3))
