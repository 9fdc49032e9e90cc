"""`models.rfdn_baseline` package (test_demo.py:26)."""
