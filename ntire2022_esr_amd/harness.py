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
import math
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


def make_synthetic_dataset(root, n, seed_image=None):
    """`--synthetic N`: N DIV2K-validation-shaped LR / HR PNG pairs for data-free boxes, laid out like select_dataset()
    expects (DIV2K_valid_LR/08xxx4.png, DIV2K_valid_HR/08xx.png).  HR = a natural image (tests/golden/test.bmp by default)
    mirror-tiled to 4H x 4W and rolled per index, LR = its PIL-bicubic x4 reduction: natural statistics, so the PNG codec
    works as hard as on DIV2K.  PSNR values on it mean nothing; wall-clock images/s of the pipeline does."""
    from PIL import Image
    shapes = [(339, 510), (339, 510), (384, 510), (339, 510), (510, 339), (294, 510), (339, 510), (345, 510), (510, 384), (339, 510)]
    if seed_image is None:
        seed_image = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "test.bmp")
    base = np.array(Image.open(seed_image).convert("RGB"))
    os.makedirs(os.path.join(root, "DIV2K_valid_LR"), exist_ok=True)
    os.makedirs(os.path.join(root, "DIV2K_valid_HR"), exist_ok=True)
    pairs = []
    for i in range(n):
        h, w = shapes[i % len(shapes)]
        hr = np.pad(base, ((0, 4 * h - base.shape[0]), (0, 4 * w - base.shape[1]), (0, 0)), mode="symmetric")
        hr = np.ascontiguousarray(np.roll(hr, (37 * i, 91 * i), axis=(0, 1)))
        lr = np.array(Image.fromarray(hr).resize((w, h), Image.BICUBIC))
        lp = os.path.join(root, f"DIV2K_valid_LR/{801 + i:04}x4.png")
        hp = os.path.join(root, f"DIV2K_valid_HR/{801 + i:04}.png")
        if not (os.path.exists(lp) and os.path.exists(hp)):
            Image.fromarray(lr).save(lp)
            Image.fromarray(hr).save(hp)
        pairs.append((lp, hp))
    return pairs


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
    every rank (lists in image order, averages in index order).

    On the GPU the loop is a three-stage pipeline -- the reference's loop is strictly serial and on DIV2K
    its PNG decode / encode (2040x1356 HR) dwarfs a few-ms forward:
      readers   a thread pool decodes the LR / HR PNGs of the next images while the GPU works (PIL's inflate drops the GIL)
      GPU       H2D of the LR (and HR, uint8), prepare(), event-bracketed forward(), tensor2uint + squared error on the
                device, asynchronous D2H of the uint8 SR image into pinned memory; nothing here waits for the host
      writers   a thread pool encodes / writes the SR PNGs
    Images retire IN ORDER from a small in-flight window, so the log lines, result lists and files are those of the serial
    loop, bit for bit; each image's runtime is its own event pair around forward(), as in the reference."""
    import time
    sf = 4
    border = sf
    rank, world = getattr(args, "rank", 0), getattr(args, "world", 1)
    data_path = pairs if pairs is not None else select_dataset(args.data_dir, mode)
    save_path = os.path.join(args.save_dir, model_name, "test" if mode == "test" else "valid")
    os.makedirs(save_path, exist_ok=True)
    use_cuda = device.type == "cuda"
    want_ssim = getattr(args, "ssim", False)
    mine = D.shard(len(data_path), rank, world)
    rows = []
    t_wall = time.perf_counter()

    def load_pair(i):
        lr_path, hr_path = data_path[i]
        return util.imread_uint(lr_path, n_channels=3), util.modcrop(util.imread_uint(hr_path, n_channels=3).squeeze(), sf)

    def log_row(i, ms, psnr, img_sr_u8, img_hr, ssim=None):
        img_name, ext = os.path.splitext(os.path.basename(data_path[i][1]))
        if want_ssim:
            if ssim is None:                     # serial loop: host evaluation; the pipeline passes the device result (ops.ssim_sum_device)
                ssim = util.calculate_ssim(img_sr_u8, img_hr, border=border)
            logger.info("{:s} - PSNR: {:.2f} dB; SSIM: {:.4f}.".format(img_name + ext, psnr, ssim))
        else:
            ssim = float("nan")
            logger.info("{:s} - PSNR: {:.2f} dB".format(img_name + ext, psnr))
        rows.append((i, ms, psnr, ssim))
        return os.path.join(save_path, img_name[:4] + ext)

    if not use_cuda or not getattr(args, "device_metrics", True):
        # serial reference loop (CPU stand-in models in the tests; device_metrics=False keeps the host metric path testable)
        if use_cuda:
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.reset_peak_memory_stats(device)
        for i in mine:
            lr, img_hr = load_pair(i)
            img_lr = util.uint2tensor4(lr, data_range).to(device)
            if use_cuda:
                if hasattr(model, "prepare"):
                    b, c, h, w = img_lr.shape
                    t = None if tile is None else min(tile, h, w)
                    model.prepare((b, c, h, w) if t is None else (b, c, t, t), device)
                start.record()
                img_sr = forward(img_lr, model, tile)
                end.record()
                torch.cuda.synchronize()
                ms = start.elapsed_time(end)
            else:
                t0 = time.perf_counter()
                img_sr = forward(img_lr, model, tile)
                ms = (time.perf_counter() - t0) * 1e3
            img_sr = util.tensor2uint(img_sr, data_range)
            psnr = util.calculate_psnr(img_sr, img_hr, border=border)
            util.imsave(img_sr, log_row(i, ms, psnr, img_sr, img_hr))
    else:
        from concurrent.futures import ThreadPoolExecutor
        from . import ops
        torch.cuda.reset_peak_memory_stats(device)
        nio = max(1, int(getattr(args, "io_workers", 8)))
        window = max(1, int(getattr(args, "inflight", 3)))
        readers, writers = ThreadPoolExecutor(nio), ThreadPoolExecutor(nio)
        copy_stream = torch.cuda.Stream(device)
        # --gpu_streams S > 1: image k's H2D, forward and metrics go to compute stream k % S (the engine keeps one workspace per
        # stream), so independent images overlap on the GPU.  Throughput only: an image's event-bracketed runtime then includes
        # the time it shared the chip, so the reference's runtime column needs the default S = 1.
        ngs = max(1, int(getattr(args, "gpu_streams", 1)))
        compute_streams = [torch.cuda.Stream(device) for _ in range(ngs)] if ngs > 1 else [torch.cuda.current_stream(device)]
        window = max(window, ngs)
        ahead = [readers.submit(load_pair, i) for i in mine[:nio + window]]
        nxt = len(ahead)
        inflight, writes = [], []

        prep_ms = []

        def enqueue(k, i, lr, img_hr, again=False):
            """H2D, prepare(), the event-bracketed forward and the device-side metrics of image i on compute stream k % ngs; returns
            its in-flight record.  Nothing here waits for the GPU.  again: the image is re-enqueued after a non-finite output ahead of
            it -- its prepare() time was recorded the first time."""
            with torch.cuda.stream(compute_streams[k % ngs]):
                img_lr = util.uint2tensor4(lr, data_range).to(device, non_blocking=True)
                hr_dev = torch.from_numpy(np.ascontiguousarray(img_hr)).to(device, non_blocking=True)
                if hasattr(model, "prepare"):
                    # plan construction / workspace zero fill are not part of the forward the reference times
                    # (test_demo.py:429-432 brackets model(img_lq) only); whole-image and tiled shapes alike
                    b, c, h, w = img_lr.shape
                    t = None if tile is None else min(tile, h, w)
                    tp = time.perf_counter()
                    model.prepare((b, c, h, w) if t is None else (b, c, t, t), device)
                    if not again:
                        prep_ms.append((time.perf_counter() - tp) * 1e3)
                start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                start.record()
                img_sr = forward(img_lr, model, tile)
                end.record()
                # Inf / NaN check folded into tensor2uint (esr_tensor2uint_u8_chk): one int on the device, no full-size isfinite pass
                bad_dev = torch.zeros(1, dtype=torch.int32, device=device)
                sr_dev = ops.tensor2uint_device(img_sr, data_range, nonfinite=bad_dev)
                if sr_dev.shape != hr_dev.shape:
                    raise ValueError('Input images must have the same dimensions.')
                se_dev = ops.sqerr_device(sr_dev, hr_dev, border=border)
                # (device scalar, count); an image whose border-cropped area is smaller than the 11x11 window has no device SSIM
                # (esr_ssim_partials == 0): the host restatement then evaluates it when the image retires, like the serial loop
                ssim_dev = None
                ssim_host = want_ssim and ops.L.lib().esr_ssim_partials(sr_dev.shape[0], sr_dev.shape[1], sr_dev.shape[2], border) == 0
                if want_ssim and not ssim_host:
                    ssim_dev = ops.ssim_sum_device(sr_dev, hr_dev, border=border)
                ready = torch.cuda.Event()
                ready.record()
                sr_host = torch.empty(sr_dev.shape, dtype=torch.uint8, pin_memory=True)
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(ready)
                    sr_host.copy_(sr_dev, non_blocking=True)
                    sr_dev.record_stream(copy_stream)
                    done = torch.cuda.Event()
                    done.record()
            hh, ww = sr_dev.shape[:2]
            return dict(k=k, i=i, lr=lr, hr=img_hr, start=start, end=end, done=done, sr_host=sr_host, se=se_dev, ssim=ssim_dev, ssim_host=ssim_host, bad=bad_dev,
                        count=(hh - 2 * border) * (ww - 2 * border) * sr_dev.shape[2])

        def retire():
            r = inflight.pop(0)
            r["done"].synchronize()                              # this image's D2H (and everything before it) has finished
            ms = r["start"].elapsed_time(r["end"])
            if bool(r["bad"].item()) and hasattr(model, "invalidate_workspaces"):
                # Inf / NaN in this output (activations overflowed, 16-bit modes).  The workspaces are shared between shapes without
                # re-zeroing, so whatever this image left in other shapes' pad slots must not reach another image: clear the
                # workspaces and RUN AGAIN the images that were enqueued behind it in the meantime (the check is read
                # when the image retires, up to `window` images late).  Their records are replaced in place: order, log lines
                # and files stay those of the serial loop.
                logger.warning("non-finite values in the output of image %d: workspaces cleared, %d image(s) in flight run again", r["i"], len(inflight))
                torch.cuda.synchronize(device)
                model.invalidate_workspaces()
                for j, q in enumerate(inflight):
                    inflight[j] = enqueue(q["k"], q["i"], q["lr"], q["hr"], again=True)
            se = int(r["se"].item())
            psnr = float("inf") if se == 0 else 20 * math.log10(255.0 / math.sqrt(se / r["count"]))
            ssim = float(r["ssim"][0].item()) / r["ssim"][1] if r["ssim"] is not None else None
            img_sr = r["sr_host"].numpy()
            if r["ssim_host"]:
                ssim = util.calculate_ssim(img_sr, r["hr"], border=border)
            writes.append(writers.submit(util.imsave, img_sr, log_row(r["i"], ms, psnr, img_sr, r["hr"], ssim)))

        for k, i in enumerate(mine):
            lr, img_hr = ahead[k].result()
            ahead[k] = None
            if nxt < len(mine):
                ahead.append(readers.submit(load_pair, mine[nxt]))
                nxt += 1
            inflight.append(enqueue(k, i, lr, img_hr))
            if len(inflight) > window:
                retire()
        while inflight:
            retire()
        for w_ in writes:
            w_.result()
        readers.shutdown()
        writers.shutdown()
        if not hasattr(args, "pipeline"):
            args.pipeline = {}
        # how this loop's columns relate to the reference's (test_demo.py:429-433, 467): reported next to the results, not in them
        args.pipeline[mode + "_measurement"] = {
            "runtime_bracket": "event pair around forward() only; prepare() (plan construction, weight repack after load_state_dict, "
                               "workspace growth / zero fill -- lazy work the reference's bracket would contain on an image's first "
                               "shape) runs before start.record(); its host time is listed here",
            "prepare_host_ms_total": round(sum(prep_ms), 3), "prepare_host_ms_max": round(max(prep_ms), 3) if prep_ms else 0.0,
            "memory": "max_memory_allocated includes the pipeline's own device buffers (the HR image, the uint8 SR image and up to "
                      f"{window} images in flight); the serial loop (device_metrics=False) keeps only the reference's tensors",
            "gpu_streams": ngs}
    wall = time.perf_counter() - t_wall
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
    # wall clock of this rank's loop (decode + forward + metrics + encode): what the pipeline is for.  Not a key of the
    # reference's result dict, so it travels on `args.pipeline` (main() writes it to pipeline.json) and in the log
    ips = len(mine) * world / wall if wall > 0 else 0.0
    if not hasattr(args, "pipeline"):
        args.pipeline = {}
    args.pipeline[mode] = {"wall_seconds": wall, "images": len(data_path), "ranks": world, "images_per_s": ips}
    logger.info("{:>16s} : {:.2f} images/s wall clock ({} images on {} rank(s), PNG decode / encode included)".format(
        "Pipeline", ips, len(data_path), world))
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
    pairs = None
    if getattr(args, "synthetic", 0):
        if rank == 0:
            make_synthetic_dataset(os.path.join(args.save_dir, "_synthetic"), args.synthetic)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        pairs = make_synthetic_dataset(os.path.join(args.save_dir, "_synthetic"), args.synthetic)   # paths only: files exist
    results[model_name] = run(model, model_name, data_range, tile, logger, device, args, mode="valid", pairs=pairs)
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
        json.dump(getattr(args, "pipeline", {}), open(os.path.join(os.getcwd(), "pipeline.json"), "w"))
        json.dump(results, open(json_path, "w"))
        open(os.path.join(os.getcwd(), "results.txt"), "w").write(results_table(results, args.include_test))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return results


def build_parser():
    p = argparse.ArgumentParser("NTIRE2022-EfficientSR")
    # (the reference's defaults are its author's cluster paths, test_demo.py:570-571; neutral ones here)
    p.add_argument("--data_dir", default="./data/NTIRE2022_Challenge", type=str,
                   help="root holding DIV2K_valid_LR / DIV2K_valid_HR (/ DIV2K_test_LR), test_demo.py:344-361")
    p.add_argument("--save_dir", default="./results", type=str)
    p.add_argument("--model_id", default=0, type=int)
    p.add_argument("--include_test", action="store_true", help="Inference on the DIV2K test set")
    p.add_argument("--ssim", action="store_true", help="Calculate SSIM")
    p.add_argument("--model_zoo", default=None, type=str, help="directory holding the reference's .pth checkpoints")
    p.add_argument("--synthetic", default=0, type=int, metavar="N",
                   help="run on N generated DIV2K-val-shaped PNG pairs under save_dir/_synthetic instead of data_dir")
    p.add_argument("--io_workers", default=8, type=int, help="PNG decode / encode threads per rank")
    p.add_argument("--inflight", default=3, type=int, help="images in flight between the GPU and the writers")
    p.add_argument("--gpu_streams", default=1, type=int,
                   help="compute streams per rank; > 1 overlaps independent images on the GPU (throughput; the per-image "
                        "runtime column then includes shared time, keep 1 for the reference's runtime semantics)")
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
