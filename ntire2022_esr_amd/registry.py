"""Model registry for the hot-path networks: the entries of `select_model` (test_demo.py:13-341)
this engine implements.  Returns (model, name, data_range, tile) like the reference; weights come from
`model_zoo/<ckpt>.pth` when a reference checkout is given, else from this repo's exported
`weights/<ckpt>.safetensors` (same tensors, key for key)."""
import os

import torch

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# model_id -> (display name, checkpoint stem, data_range, tile, ctor)
def _entries():
    import contextlib
    import io
    from .bsrn import BSRN
    from .imdn import IMDN
    from .rfdn import RFDN
    from .rlfn import RLFN_cut
    return {
        -1: ("IMDN_baseline", "imdn_baseline", 1.0, None, lambda: IMDN(in_nc=3, out_nc=3, nc=64, nb=8, upscale=4)),
        0: ("RFDN_baseline", "rfdn_baseline", 255.0, None, lambda: RFDN()),                    # test_demo.py:24-30
        4: ("RLFN", "team04_rlfn", 255.0, None, lambda: RLFN_cut(in_nc=3, out_nc=3)),        # test_demo.py:52-58
        # free riders: same graphs, own checkpoints (test_demo.py:66-72, 175-181, 203-209)
        6: ("V1", "team06_v1", 1.0, None, lambda: RFDN(in_nc=3, nf=50, num_modules=4, out_nc=3, upscale=4)),
        22: ("RFDN40", "team22_rep_rfdn", 1.0, None, lambda: RFDN(in_nc=3, nf=40, num_modules=4, out_nc=3, upscale=4)),
        # near riders: the RFDN graph with two switches (test_demo.py:76-82, 302-308)
        8: ("RFDN", "team08_sfdn", 1.0, None, lambda: RFDN(block_residual=False, esa_conv_f=False)),
        40: ("RFDNPrune", "team40_rfdn_pruned", 255.0, None,
             lambda: RFDN(in_nc=3, nf=40, num_modules=4, out_nc=3, upscale=4, block_residual=False, esa_f=12)),
        26: ("IMDN", "team26_imdn_nb7", 1.0, None, lambda: IMDN(in_nc=3, out_nc=3, nc=64, nb=7, upscale=4, act_mode='L',
                                                                upsample_mode='pixelshuffle')),
        18: ("RFDNFINALB5", "team18_bsrn", 1.0, None,                                                 # test_demo.py:150-157
             lambda: BSRN(num_in_ch=3, num_feat=48, num_block=5, num_out_ch=3, upscale=4, conv='BSConvU',
                          upsampler='pixelshuffledirect')),
    }


def supported_ids():
    return sorted(_entries())


def load_checkpoint(stem, model_zoo=None):
    """state_dict for `stem`; unwraps the containers the reference unwraps (test_demo.py:157 'params')."""
    if model_zoo is not None:
        for ext in (".pth", ".pt"):
            p = os.path.join(model_zoo, stem + ext)
            if os.path.exists(p):
                sd = torch.load(p, map_location="cpu", weights_only=True)    # plain tensors only: no pickle code execution
                return sd["params"] if isinstance(sd, dict) and "params" in sd and len(sd) == 1 else sd
    from safetensors.torch import load_file
    return load_file(os.path.join(_REPO, "weights", stem + ".safetensors"))


def select_model(model_id, device, model_zoo=None):
    ent = _entries().get(model_id)
    if ent is None:
        raise NotImplementedError(f"Model {model_id} is not implemented.")       # test_demo.py:333
    disp, stem, data_range, tile, ctor = ent
    name = f"{model_id:02}_{disp}"
    model = ctor()
    model.load_state_dict(load_checkpoint(stem, model_zoo), strict=True)
    model.eval()
    for _, v in model.named_parameters():
        v.requires_grad = False
    return model.to(device), name, data_range, tile
