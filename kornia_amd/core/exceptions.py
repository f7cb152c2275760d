"""Validation exception types.

Mirror of the reference's exception hierarchy (kornia/core/exceptions.py:33-126) so that
``pytest.raises(ShapeError)``-style parity tests read the same.  When a real ``kornia`` is
importable its classes are re-used, so a patched Kornia raises Kornia's own types.
"""
from __future__ import annotations

import sys
from typing import Any

__all__ = ["BaseError", "DeviceError", "ShapeError", "TypeCheckError", "ValueCheckError"]


def _from_kornia():
    mod = sys.modules.get("kornia.core.exceptions")
    if mod is None:
        return None
    try:
        return mod.BaseError, mod.ShapeError, mod.TypeCheckError, mod.ValueCheckError, mod.DeviceError
    except AttributeError:
        return None


def _detail_error(name: str, fields: tuple, doc: str, base: type) -> type:
    """An error type that carries optional keyword-only details (``None`` when not given), e.g. ``ShapeError(msg, actual_shape=...)``."""

    def __init__(self, message: str, **details: Any) -> None:
        unknown = sorted(set(details) - set(fields))
        if unknown:
            raise TypeError(f"{name}() got unexpected keyword arguments {unknown}")
        base.__init__(self, message)
        for field in fields:
            setattr(self, field, details.get(field))

    return type(name, (base,), {"__init__": __init__, "__doc__": doc, "__module__": __name__, "_fields": fields})


_k = _from_kornia()
if _k is not None:
    BaseError, ShapeError, TypeCheckError, ValueCheckError, DeviceError = _k
else:

    class BaseError(Exception):
        """Root of all validation errors."""

    ShapeError = _detail_error("ShapeError", ("actual_shape", "expected_shape"), "A tensor does not have the required shape.", BaseError)
    TypeCheckError = _detail_error("TypeCheckError", ("actual_type", "expected_type"), "An argument has the wrong type.", BaseError)
    ValueCheckError = _detail_error("ValueCheckError", ("actual_value", "expected_range"), "A value lies outside its allowed range.", BaseError)
    DeviceError = _detail_error("DeviceError", ("actual_devices", "expected_device"), "Tensors live on different devices.", BaseError)
