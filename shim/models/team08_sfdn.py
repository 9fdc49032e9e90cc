"""`from models.team08_sfdn import RFDN` (test_demo.py:76-82): the RFDN graph without the in-block residuals and without ESA's conv_f."""
from ntire2022_esr_amd.rfdn import RFDN as _RFDN


class RFDN(_RFDN):
    def __init__(self, conv=None, in_nc=3, nf=50, num_modules=4, out_nc=3, upscale=4):
        # `conv` (the reference's default_conv factory) selects nothing here: the kernels implement that one convolution
        super().__init__(in_nc=in_nc, nf=nf, num_modules=num_modules, out_nc=out_nc, upscale=upscale,
                         block_residual=False, esa_conv_f=False)


__all__ = ["RFDN"]
