"""`from models.team06_v1 import v1` (test_demo.py:66-72): the rfdn_baseline graph with its own checkpoint."""
from ntire2022_esr_amd.rfdn import RFDN as v1  # noqa: F401

__all__ = ["v1"]
