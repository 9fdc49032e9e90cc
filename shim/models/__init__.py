"""`models` package of an unmodified test_demo.py, backed by ntire2022_esr_amd (see shim/README.md)."""
