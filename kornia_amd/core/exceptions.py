"""Validation exception types.

Mirror of the reference's exception hierarchy (kornia/core/exceptions.py:33-126) so that
``pytest.raises(ShapeError)``-style parity tests read the same.  When a real ``kornia`` is
importable its classes are re-used, so a patched Kornia raises Kornia's own types.
"""
from __future__ import annotations

import sys
from typing import Any, Optional

__all__ = ["BaseError", "DeviceError", "ShapeError", "TypeCheckError", "ValueCheckError"]


def _from_kornia():
    mod = sys.modules.get("kornia.core.exceptions")
    if mod is None:
        return None
    try:
        return mod.BaseError, mod.ShapeError, mod.TypeCheckError, mod.ValueCheckError, mod.DeviceError
    except AttributeError:
        return None


_k = _from_kornia()
if _k is not None:
    BaseError, ShapeError, TypeCheckError, ValueCheckError, DeviceError = _k
else:

    class BaseError(Exception):
        """Root of all validation errors."""

    class ShapeError(BaseError):
        def __init__(self, message: str, *, actual_shape=None, expected_shape=None):
            super().__init__(message)
            self.actual_shape = actual_shape
            self.expected_shape = expected_shape

    class TypeCheckError(BaseError):
        def __init__(self, message: str, *, actual_type: Optional[type] = None, expected_type: Any = None):
            super().__init__(message)
            self.actual_type = actual_type
            self.expected_type = expected_type

    class ValueCheckError(BaseError):
        def __init__(self, message: str, *, actual_value: Any = None, expected_range: Any = None):
            super().__init__(message)
            self.actual_value = actual_value
            self.expected_range = expected_range

    class DeviceError(BaseError):
        def __init__(self, message: str, *, actual_devices: Optional[list] = None, expected_device: Any = None):
            super().__init__(message)
            self.actual_devices = actual_devices
            self.expected_device = expected_device
