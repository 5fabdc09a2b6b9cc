"""Batched UNet pass (pxt_unet_forward_batch) vs batch size: ms per call, per image pair, TFLOP/s.
    python scripts/bench_unet_batch.py [--sizes 1,2,4,8,16] [--per-image-plan] [--hw 480x640] [--reps 10]"""
import argparse
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pixtrack_amd.unet import UNet, make_synthetic_unet_weights
sys.path.insert(0, str(Path(__file__).resolve().parent))
from bench_unet import flops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1,2,4,8,16")
    ap.add_argument("--per-image-plan", action="store_true")
    ap.add_argument("--hw", default="480x640")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    H, W = (int(x) for x in args.hw.split("x"))
    dev = torch.device("cuda:0")
    net = UNet(make_synthetic_unet_weights(7), dev)
    net.set_batch_plan(args.per_image_plan)
    g = torch.Generator(device="cpu").manual_seed(3)
    for B in (int(x) for x in args.sizes.split(",")):
        items = []
        for i in range(B):
            if i % 2 == 0:  # a reference render: uint8, no mask
                items.append(((torch.rand(H, W, 3, generator=g) * 255).to(torch.uint8).to(dev), None, False))
            else:           # a masked query: float32 + silhouette mask
                m = torch.zeros(H, W, dtype=torch.uint8)
                m[H // 4:3 * H // 4, W // 4:3 * W // 4] = 1
                items.append(((torch.rand(H, W, 3, generator=g) * 255).to(dev), m.to(dev), True))
        for _ in range(3):
            net.forward_packed_batch(items)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            net.forward_packed_batch(items)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        print(f"B={B:2d} {W}x{H} plan={'per-image' if args.per_image_plan else 'batch'}: {ms:7.3f} ms/call {ms / B * 2:6.3f} ms/pair "
              f"{flops(H, W) * B / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
