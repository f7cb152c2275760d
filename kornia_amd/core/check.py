"""Argument validators with the reference's names, semantics and messages.

Reference: kornia/core/check.py:63-128 (global on/off switch, ``KORNIA_CHECKS`` env, ``python -O``),
:131-216 (KORNIA_CHECK_SHAPE), :219-254 (KORNIA_CHECK), :323-370 (KORNIA_CHECK_IS_TENSOR).
"""
from __future__ import annotations

import os
from typing import Any, Optional

import torch

from .exceptions import BaseError, ShapeError, TypeCheckError

__all__ = [
    "KORNIA_CHECK",
    "KORNIA_CHECK_IS_TENSOR",
    "KORNIA_CHECK_SHAPE",
    "are_checks_enabled",
    "device_value_checks",
    "set_device_value_checks",
    "disable_checks",
    "enable_checks",
]


def _initial_state() -> bool:
    env = os.getenv("KORNIA_CHECKS")
    if env is not None:
        return env.lower() in ("1", "true", "yes", "on")
    return __debug__


_ENABLED: bool = _initial_state()


def are_checks_enabled() -> bool:
    return _ENABLED


def disable_checks() -> None:
    global _ENABLED
    _ENABLED = False


def enable_checks() -> None:
    global _ENABLED
    _ENABLED = True


# Checks on the VALUES of device tensors (e.g. ``(sigma > 0).all()`` in gaussian_blur2d, kornia/filters/gaussian.py:103-108)
# need a device -> host read that drains the stream.  The native path does not make them by default;
# KORNIA_AMD_DEVICE_VALUE_CHECKS=1 or set_device_value_checks(True) restores the reference's behaviour.
_DEVICE_VALUE_CHECKS: bool = os.getenv("KORNIA_AMD_DEVICE_VALUE_CHECKS", "0").lower() in ("1", "true", "yes", "on")


def device_value_checks() -> bool:
    return _DEVICE_VALUE_CHECKS and _ENABLED


def set_device_value_checks(enabled: bool) -> bool:
    """Returns the previous setting."""
    global _DEVICE_VALUE_CHECKS
    old, _DEVICE_VALUE_CHECKS = _DEVICE_VALUE_CHECKS, bool(enabled)
    return old


def _fail_shape(x: torch.Tensor, shape: list[str], head: str, msg: Optional[str]):
    actual = list(x.shape)
    text = f"{head}\n  Expected shape: {shape}\n  Actual shape: {actual}"
    if msg is not None:
        text += f"\n  {msg}"
    raise ShapeError(text, actual_shape=actual, expected_shape=shape)


def KORNIA_CHECK_SHAPE(x: torch.Tensor, shape: list[str], msg: Optional[str] = None, raises: bool = True) -> bool:
    """``shape`` entries are names (any size) or numerals (exact size); a leading or trailing
    ``"*"`` matches any number of extra dimensions."""
    if not _ENABLED:
        return True
    if shape[0] == "*":
        want, got = shape[1:], x.shape[-len(shape) + 1 :]
    elif shape[-1] == "*":
        want, got = shape[:-1], x.shape[: len(shape) - 1]
    else:
        want, got = shape, x.shape
    if len(got) != len(want):
        if raises:
            _fail_shape(x, shape, f"Shape dimension mismatch: expected {len(want)} dimensions, got {len(got)}.", msg)
        return False
    for i, name in enumerate(want):
        if name.isnumeric() and got[i] != int(name):
            if raises:
                _fail_shape(x, shape, f"Shape mismatch at dimension {i}: expected {int(name)}, got {got[i]}.", msg)
            return False
    return True


def KORNIA_CHECK(condition: bool, msg: Optional[str] = None, raises: bool = True) -> bool:
    if not _ENABLED:
        return True
    if not condition:
        if raises:
            raise BaseError("Validation condition failed" if msg is None else msg)
        return False
    return True


def KORNIA_CHECK_IS_TENSOR(x: Any, msg: Optional[str] = None, raises: bool = True) -> bool:
    if not _ENABLED:
        return True
    if not isinstance(x, torch.Tensor):
        if raises:
            text = f"Type mismatch: expected Tensor, got {type(x)!s}."
            if msg is not None:
                text += f"\n  {msg}"
            raise TypeCheckError(text, actual_type=type(x), expected_type=torch.Tensor)
        return False
    return True
