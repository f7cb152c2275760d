from .check import (
    KORNIA_CHECK,
    KORNIA_CHECK_IS_TENSOR,
    KORNIA_CHECK_SHAPE,
    are_checks_enabled,
    disable_checks,
    enable_checks,
)
from .exceptions import BaseError, DeviceError, ShapeError, TypeCheckError, ValueCheckError
