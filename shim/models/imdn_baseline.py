"""`from models.imdn_baseline import IMDN` (test_demo.py:20, 205) -> the HIP-engine IMDN (same ctor keywords, same 86 keys)."""
from ntire2022_esr_amd.imdn import IMDN  # noqa: F401

__all__ = ["IMDN"]
