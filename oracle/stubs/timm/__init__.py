"""Test-only stub of `timm` (only trunc_normal_ is used by the reference)."""
