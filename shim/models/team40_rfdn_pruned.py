"""`from models.team40_rfdn_pruned import RFDN as RFDNPrune` (test_demo.py:302-308): RFDN nf = 40 without the in-block residuals, ESA width 12."""
from ntire2022_esr_amd.rfdn import RFDN as _RFDN


class RFDN(_RFDN):
    def __init__(self, in_nc=3, nf=40, num_modules=4, out_nc=3, upscale=4):
        super().__init__(in_nc=in_nc, nf=nf, num_modules=num_modules, out_nc=out_nc, upscale=upscale,
                         block_residual=False, esa_f=12)


__all__ = ["RFDN"]
