"""`from models.team04_rlfn import RLFN_cut` (test_demo.py:54) -> the HIP-engine RLFN_cut."""
from ntire2022_esr_amd.rlfn import RLFN_cut  # noqa: F401

__all__ = ["RLFN_cut"]
