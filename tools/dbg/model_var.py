"""IMDN / other model forward timing with a library variant: model_var.py <variant|prod> [model ids]"""
import os, sys
here = os.path.dirname(os.path.abspath(__file__)); root = os.path.dirname(os.path.dirname(here))
sys.path.insert(0, root)
from ntire2022_esr_amd import _lib as L
var = sys.argv[1]
if var != "prod": L.SO_PATH = os.path.join(here, f"libesr_var_{var}.so")
sys.argv = [sys.argv[0]] + sys.argv[2:]
exec(open(os.path.join(root, "tools/quick_time.py")).read())
