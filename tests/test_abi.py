"""CPU: the C-ABI library loads and exports every function include/pm_mi355x.h declares (no compute)."""
import ctypes
import os

from pytorch_models import _hip


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_hip.LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    L = ctypes.CDLL(_hip.LIB_PATH)
    declared = _hip.header_functions()
    assert declared, "no functions parsed from the header"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/pm_mi355x.h but not exported"
    assert sorted(_hip.SIGNATURES) == declared, "ctypes SIGNATURES table out of sync with the header"


def test_abi_version_and_error_strings():
    L = _hip.lib()
    assert L.pm_abi_version() == 1
    for code in range(0, 5):
        assert L.pm_strerror(code)
    assert b"unknown" in L.pm_strerror(99)


def test_host_side_validation_launches_nothing():
    """Rejected calls return an error code before any HIP call, so this is safe without a GPU."""
    L = _hip.lib()
    assert L.pm_linear_bf16(None, 0, None, 0, None, None, 0, 0, None, 0, 0, 1, 1, 1, 0, None) == 1  # PM_EINVAL
    buf = ctypes.create_string_buffer(4096)
    p = ctypes.addressof(buf) // 16 * 16 + 16
    # K not a multiple of 8 -> PM_EUNSUPPORTED
    assert L.pm_linear_bf16(p, 72, p, 72, None, None, 0, 0, p, 8, 0, 4, 8, 68, 0, None) == 2
    # misaligned leading dimension -> PM_EALIGN
    assert L.pm_linear_bf16(p, 65, p, 64, None, None, 0, 0, p, 8, 0, 4, 8, 64, 0, None) == 4
    assert L.pm_layernorm(p, 12, 0, p, p, 1e-5, p, 12, 0, 4, 12, None) == 2
    assert L.pm_vit_tokens(p, p, p, p, None, p, 1, 224, 224, 14, 384, None) == 2
