"""MI355X-native engine for the NTIRE2022_ESR test_demo.py forward path.

Package contents (only what the path needs):
  csrc/        HIP kernels + C ABI (include/esr_hip.h) -> libesr_hip.so
  _lib.py      ctypes binding (fails loudly, no fallback)
  engine.py    weight packing, workspace, op-list replay
  imdn.py ...  drop-in nn.Modules with the reference's ctor / state_dict surface
"""
from .bsrn import BSRN  # noqa: F401
from .imdn import IMDN  # noqa: F401
from .rfdn import RFDN  # noqa: F401
from .rlfn import RLFN_cut  # noqa: F401

__all__ = ["BSRN", "IMDN", "RFDN", "RLFN_cut"]
