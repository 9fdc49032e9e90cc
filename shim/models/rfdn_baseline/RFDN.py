"""`from models.rfdn_baseline.RFDN import RFDN` (test_demo.py:26) -> the HIP-engine RFDN (same ctor keywords, same 128 keys)."""
from ntire2022_esr_amd.rfdn import RFDN  # noqa: F401

__all__ = ["RFDN"]
