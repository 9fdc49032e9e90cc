"""Evaluation loop around the forward path: the engine's own `test_demo.py`.

Mirrors forward() (test_demo.py:364-391), select_dataset() (:344-361), run() (:394-477) and main()
(:480-563): same five CLI flags, same per-image log line, same results.json keys and results.txt
columns -- plus image-level sharding over the GPUs of a node (one process per GPU, launched with
torch.distributed.run) and a `--synthetic N` mode for data-free boxes.

  python -m ntire2022_esr_amd.harness --data_dir D --save_dir S --model_id -1 [--include_test] [--ssim]
"""
import argparse
import json
import logging
import os
import sys

import numpy as np
import torch

from . import dist as D
from . import image_util as util
from .registry import select_model
from .summary import model_complexity


def select_dataset(data_dir, mode):
    if mode == "test":
        return [(os.path.join(data_dir, f"DIV2K_test_LR/{i:04}.png"),
                 os.path.join(data_dir, f"DIV2K_test_HR/{i:04}.png")) for i in range(901, 1001)]
    return [(os.path.join(data_dir, f"DIV2K_valid_LR/{i:04}x4.png"),
             os.path.join(data_dir, f"DIV2K_valid_HR/{i:04}.png")) for i in range(801, 901)]


def forward(img_lq, model, tile=None, tile_overlap=32, scale=4):
    """Whole image, or overlap-tiled accumulate / normalise (stride = tile - overlap)."""
    if tile is None:
        return model(img_lq)
    b, c, h, w = img_lq.size()
    tile = min(tile, h, w)
    stride = tile - tile_overlap
    hs = list(range(0, h - tile, stride)) + [h - tile]
    ws = list(range(0, w - tile, stride)) + [w - tile]
    acc = torch.zeros(b, c, h * scale, w * scale).type_as(img_lq)
    hit = torch.zeros_like(acc)
    for y in hs:
        for x in ws:
            out = model(img_lq[..., y:y + tile, x:x + tile].contiguous())
            acc[..., y * scale:(y + tile) * scale, x * scale:(x + tile) * scale].add_(out)
            hit[..., y * scale:(y + tile) * scale, x * scale:(x + tile) * scale].add_(1.0)
    return acc.div_(hit)


def run(model, model_name, data_range, tile, logger, device, args, mode="test", pairs=None, timer=None):
    """Per-image loop (test_demo.py:416-465) over this rank's shard; returns the reference's result dict on
    every rank (lists in image order, averages in index order)."""
    sf = 4
    border = sf
    rank, world = getattr(args, "rank", 0), getattr(args, "world", 1)
    data_path = pairs if pairs is not None else select_dataset(args.data_dir, mode)
    save_path = os.path.join(args.save_dir, model_name, "test" if mode == "test" else "valid")
    os.makedirs(save_path, exist_ok=True)
    use_cuda = device.type == "cuda"
    if use_cuda:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.reset_peak_memory_stats(device)
    rows = []
    for i in D.shard(len(data_path), rank, world):
        lr_path, hr_path = data_path[i]
        img_name, ext = os.path.splitext(os.path.basename(hr_path))
        img_lr = util.uint2tensor4(util.imread_uint(lr_path, n_channels=3), data_range).to(device)
        if use_cuda:
            if hasattr(model, "prepare"):
                # plan construction / workspace zero fill are not part of the forward the reference times
                # (test_demo.py:429-432 brackets model(img_lq) only); whole-image and tiled shapes alike
                b, c, h, w = img_lr.shape
                t = None if tile is None else min(tile, h, w)
                model.prepare((b, c, h, w) if t is None else (b, c, t, t), device)
            start.record()
            img_sr = forward(img_lr, model, tile)
            end.record()
            torch.cuda.synchronize()
            ms = start.elapsed_time(end)
        else:
            import time
            t0 = time.perf_counter()
            img_sr = forward(img_lr, model, tile)
            ms = (time.perf_counter() - t0) * 1e3
        img_hr = util.modcrop(util.imread_uint(hr_path, n_channels=3).squeeze(), sf)
        if use_cuda and getattr(args, "device_metrics", True):
            # uint8 conversion and the squared-error sum run on the GPU: the SR image crosses PCIe once as uint8
            # (needed for imsave) and the PSNR as one integer
            from . import ops
            sr_dev = ops.tensor2uint_device(img_sr, data_range)
            psnr = ops.psnr_device(sr_dev, torch.from_numpy(np.ascontiguousarray(img_hr)).to(device), border=border)
            img_sr = sr_dev.cpu().numpy()
        else:
            img_sr = util.tensor2uint(img_sr, data_range)
            psnr = util.calculate_psnr(img_sr, img_hr, border=border)
        if getattr(args, "ssim", False):
            ssim = util.calculate_ssim(img_sr, img_hr, border=border)
            logger.info("{:s} - PSNR: {:.2f} dB; SSIM: {:.4f}.".format(img_name + ext, psnr, ssim))
        else:
            ssim = float("nan")
            logger.info("{:s} - PSNR: {:.2f} dB".format(img_name + ext, psnr))
        rows.append((i, ms, psnr, ssim))
        util.imsave(img_sr, os.path.join(save_path, img_name[:4] + ext))
    allrows = D.gather_rows(rows, len(data_path), rank, world, device)
    results = {f"{mode}_runtime": [float(v) for v in allrows[:, 1]],
               f"{mode}_psnr": [float(v) for v in allrows[:, 2]]}
    mem = torch.cuda.max_memory_allocated(device) / 1024 ** 2 if use_cuda else 0.0
    if world > 1 and use_cuda:
        import torch.distributed as dist
        t = torch.tensor([mem], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        mem = float(t.item())
    results[f"{mode}_memory"] = mem
    results[f"{mode}_ave_runtime"] = D.ordered_mean(results[f"{mode}_runtime"])
    results[f"{mode}_ave_psnr"] = D.ordered_mean(results[f"{mode}_psnr"])
    if getattr(args, "ssim", False):
        results[f"{mode}_ssim"] = [float(v) for v in allrows[:, 3]]
        results[f"{mode}_ave_ssim"] = D.ordered_mean(results[f"{mode}_ssim"])
    logger.info("{:>16s} : {:<.3f} [M]".format("Max Memery", results[f"{mode}_memory"]))
    logger.info("------> Average runtime of ({}) is : {:.6f} seconds".format(
        "test" if mode == "test" else "valid", results[f"{mode}_ave_runtime"]))
    return results


def results_table(results, include_test):
    """results.txt (test_demo.py:539-563)."""
    if include_test:
        fmt = "{:20s}\t{:10s}\t{:10s}\t{:14s}\t{:14s}\t{:14s}\t{:10s}\t{:10s}\t{:8s}\t{:8s}\t{:8s}\n"
        s = fmt.format("Model", "Val PSNR", "Test PSNR", "Val Time [ms]", "Test Time [ms]", "Ave Time [ms]",
                       "Params [M]", "FLOPs [G]", "Acts [M]", "Mem [M]", "Conv")
    else:
        fmt = "{:20s}\t{:10s}\t{:14s}\t{:10s}\t{:10s}\t{:8s}\t{:8s}\t{:8s}\n"
        s = fmt.format("Model", "Val PSNR", "Val Time [ms]", "Params [M]", "FLOPs [G]", "Acts [M]", "Mem [M]", "Conv")
    for k, v in results.items():
        cols = dict(val_psnr=f"{v['valid_ave_psnr']:2.2f}", val_time=f"{v['valid_ave_runtime']:3.2f}",
                    num_param=f"{v['num_parameters']:2.3f}", flops=f"{v['flops']:2.2f}",
                    acts=f"{v['activations']:2.2f}", mem=f"{v['valid_memory']:2.2f}", conv=f"{v['num_conv']:4d}")
        if include_test:
            s += fmt.format(k, cols["val_psnr"], f"{v['test_ave_psnr']:2.2f}", cols["val_time"],
                            f"{v['test_ave_runtime']:3.2f}",
                            f"{(v['valid_ave_runtime'] + v['test_ave_runtime']) / 2:3.2f}",
                            cols["num_param"], cols["flops"], cols["acts"], cols["mem"], cols["conv"])
        else:
            s += fmt.format(k, cols["val_psnr"], cols["val_time"], cols["num_param"], cols["flops"],
                            cols["acts"], cols["mem"], cols["conv"])
    return s


def main(args):
    logger = logging.getLogger("NTIRE2022-EfficientSR")
    if not logger.handlers:
        logger.setLevel(logging.INFO)
        fmt = logging.Formatter("%(asctime)s.%(msecs)03d : %(message)s", datefmt="%y-%m-%d %H:%M:%S")
        for h in (logging.FileHandler("NTIRE2022-EfficientSR.log", mode="a"), logging.StreamHandler()):
            h.setFormatter(fmt)
            logger.addHandler(h)
    if not torch.cuda.is_available():
        raise SystemExit("the HIP engine needs an MI355X; there is no CPU fallback (use oracle/ for CPU checks)")
    rank, world, local_rank = D.init_from_env(use_cuda=True)
    args.rank, args.world = rank, world
    device = torch.device("cuda", local_rank)
    if rank != 0:
        logger.setLevel(logging.WARNING)
    json_path = os.path.join(os.getcwd(), "results.json")
    results = json.load(open(json_path)) if os.path.exists(json_path) else dict()
    model, model_name, data_range, tile = select_model(args.model_id, device, getattr(args, "model_zoo", None))
    logger.info(model_name)
    results[model_name] = run(model, model_name, data_range, tile, logger, device, args, mode="valid")
    if args.include_test:
        results[model_name].update(run(model, model_name, data_range, tile, logger, device, args, mode="test"))
    c = model_complexity(model, (3, 256, 256))
    activations, flops, num_parameters = c["activations"] / 10 ** 6, c["flops"] / 10 ** 9, c["num_parameters"] / 10 ** 6
    logger.info("{:>16s} : {:<.4f} [M]".format("#Activations", activations))
    logger.info("{:>16s} : {:<d}".format("#Conv2d", c["num_conv"]))
    logger.info("{:>16s} : {:<.4f} [G]".format("FLOPs", flops))
    logger.info("{:>16s} : {:<.4f} [M]".format("#Params", num_parameters))
    results[model_name].update({"activations": activations, "num_conv": c["num_conv"], "flops": flops,
                                "num_parameters": num_parameters})
    if rank == 0:
        json.dump(results, open(json_path, "w"))
        open(os.path.join(os.getcwd(), "results.txt"), "w").write(results_table(results, args.include_test))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return results


def build_parser():
    p = argparse.ArgumentParser("NTIRE2022-EfficientSR")
    p.add_argument("--data_dir", default="/cluster/work/cvl/yawli/data/NTIRE2022_Challenge", type=str)
    p.add_argument("--save_dir", default="/cluster/work/cvl/yawli/data/NTIRE2022_Challenge/results", type=str)
    p.add_argument("--model_id", default=0, type=int)
    p.add_argument("--include_test", action="store_true", help="Inference on the DIV2K test set")
    p.add_argument("--ssim", action="store_true", help="Calculate SSIM")
    p.add_argument("--model_zoo", default=None, type=str, help="directory holding the reference's .pth checkpoints")
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
