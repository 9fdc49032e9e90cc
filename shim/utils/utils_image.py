"""`from utils import utils_image as util` (test_demo.py:10): the helpers run() calls, from ntire2022_esr_amd.image_util."""
from ntire2022_esr_amd.image_util import (calculate_psnr, calculate_ssim, imread_uint, imsave, mkdir, mkdirs, modcrop,  # noqa: F401
                                          tensor2uint, uint2tensor4)

__all__ = ["calculate_psnr", "calculate_ssim", "imread_uint", "imsave", "mkdir", "mkdirs", "modcrop", "tensor2uint", "uint2tensor4"]
