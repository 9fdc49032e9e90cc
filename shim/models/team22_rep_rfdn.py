"""`from models.team22_rep_rfdn import RFDN40` (test_demo.py:175-181): RFDN with nf = 40."""
from ntire2022_esr_amd.rfdn import RFDN


class RFDN40(RFDN):
    def __init__(self, in_nc=3, nf=40, num_modules=4, out_nc=3, upscale=4):
        super().__init__(in_nc=in_nc, nf=nf, num_modules=num_modules, out_nc=out_nc, upscale=upscale)


__all__ = ["RFDN40"]
