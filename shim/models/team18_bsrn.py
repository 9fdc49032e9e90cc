"""`from models.team18_bsrn import BSRN` (test_demo.py:152) -> the HIP-engine BSRN."""
from ntire2022_esr_amd.bsrn import BSRN  # noqa: F401

__all__ = ["BSRN"]
