"""TEST INFRASTRUCTURE ONLY - import shim for the *real* reference (kornia @ /root/reference).

Only usable inside the build container (``/root/reference`` does not exist on the GPU box).
It is used by ``oracle/make_golden.py`` to generate the committed fixtures under
``tests/golden/`` and by ``tests/test_patch_reference.py`` (auto-skipped when the
reference tree is absent).  Nothing in the product path (``kornia_amd/``) imports this.

The reference needs Python >= 3.11 (``enum.StrEnum`` kornia/config.py:20, ``enum.member``
kornia/losses/mutual_information.py:19) and the Rust wheel ``kornia_rs`` (kornia/io/io.py:24,
image file I/O only - not on the hot path); both are stubbed here.
"""
from __future__ import annotations

import enum
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("KORNIA_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "kornia"))


def import_reference():
    """Return the reference ``kornia`` module (imported from REFERENCE_ROOT, read-only tree)."""
    if not reference_available():
        raise ImportError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    if not hasattr(enum, "StrEnum"):

        class StrEnum(str, enum.Enum):
            def __str__(self) -> str:
                return str(self.value)

            @staticmethod
            def _generate_next_value_(name, start, count, last_values):
                return name.lower()

        enum.StrEnum = StrEnum
    if not hasattr(enum, "member"):
        enum.member = lambda x: x
        enum.nonmember = lambda x: x
    if "kornia_rs" not in sys.modules:
        stub = types.ModuleType("kornia_rs")
        stub.__version__ = "0.0.0-stub"
        sys.modules["kornia_rs"] = stub
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import kornia  # noqa: PLC0415

    return kornia
