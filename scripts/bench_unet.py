"""Per-image time of the HIP UNet and achieved MFMA rate (241.4 GFLOP @ 640x480)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pixtrack_amd.unet import UNet, make_synthetic_unet_weights, conv_layer_dims

def flops(H, W):
    hs = [(H >> i, W >> i) for i in range(5)]
    res = [0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 3, 2, 1, 0]
    tot = 0
    for (cin, cout), r in zip(conv_layer_dims(), res):
        tot += 2 * 9 * cin * cout * hs[r][0] * hs[r][1]
    return tot

def main():
    dev = torch.device("cuda:0")
    net = UNet(make_synthetic_unet_weights(7), dev)
    for (H, W) in [(240, 320), (480, 640), (576, 1024)]:
        img = torch.rand(H, W, 3, device=dev) * 255
        for _ in range(3):
            net.forward_packed(img, None, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            net.forward_packed(img, None, True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{W}x{H}: {ms:.3f} ms/image  {flops(H, W)/ms/1e9:.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    main()
