"""Host-side operand checks of the ctypes wrappers (no GPU needed): one device for all operands, and that device current."""
import types

import pytest
import torch

from pytorch_models._hip import ops


def fake(index):
    return types.SimpleNamespace(is_cuda=True, device=torch.device("cuda", index))


def test_cpu_tensor_is_refused():
    with pytest.raises(RuntimeError, match="HIP devices only"):
        ops.check_devices(torch.zeros(2))


def test_mixed_devices_are_refused(monkeypatch):
    monkeypatch.setattr(ops, "_current_device_index", lambda: 0)
    with pytest.raises(RuntimeError, match="different devices"):
        ops.check_devices(fake(0), None, fake(1))


def test_operands_off_the_current_device_are_refused(monkeypatch):
    monkeypatch.setattr(ops, "_current_device_index", lambda: 0)
    ops.check_devices(fake(0), fake(0))  # fine
    with pytest.raises(RuntimeError, match="current device is cuda:0"):
        ops.check_devices(fake(1), fake(1))
    monkeypatch.setattr(ops, "_current_device_index", lambda: 1)
    ops.check_devices(fake(1), None)
