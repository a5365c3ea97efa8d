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


def test_library_exports_nothing_the_header_does_not_declare():
    """exported pm_* == declared: the shipped library carries no undeclared entry point (an experiment's bench export
    rode along).  The product library is checked against pm_mi355x.h; an experiments build (PM_MI355X_LIB=...libpm_mi355x_exp.so)
    additionally against pm_mi355x_experiments.h."""
    import re
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", _hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({l.split()[-1] for l in out.splitlines() if re.match(r"^[0-9a-f]+ T pm_[a-z0-9_]+$", l.strip())})
    allowed = set(_hip.header_functions())
    if os.path.basename(_hip.LIB_PATH) != "libpm_mi355x.so":
        allowed |= set(_hip.EXPERIMENT_SIGNATURES)
    extra = [n for n in exported if n not in allowed]
    assert exported and not extra, f"exported but not declared: {extra}"


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
