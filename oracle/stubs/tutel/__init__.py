"""Test-only stub of the third-party `tutel` package (see ../README.md)."""
from .impls import communicate as net  # noqa: F401
