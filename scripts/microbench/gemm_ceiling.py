"""What the library GEMM (hipBLASLt through torch.matmul, fp16 in / fp32 accumulate) sustains on this chip for the UNet
layers' implicit-GEMM shapes (M = Cout, N = pixels of 1 / 16 images, K = 9 Cin) and for a large square: the practical MFMA
ceiling the hand-written convolution loop (~1.0 PFLOP/s on chip-filling layers) is compared with.  Not a
product path: a measurement of the machine."""
import torch


def tflops(M, N, K, iters=30):
    a = torch.randn(M, K, device="cuda", dtype=torch.float16)
    b = torch.randn(K, N, device="cuda", dtype=torch.float16)
    for _ in range(5):
        torch.matmul(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(a, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * M * N * K / ms / 1e9, ms


def main():
    print("shape (M = Cout, N = pixels, K = 9 Cin): TFLOP/s, us")
    for name, M, N, K in (
        ("square 8192", 8192, 8192, 8192),
        ("square 4096", 4096, 4096, 4096),
        ("conv2_2 240x320 x1  128->128", 128, 76800, 1152),
        ("conv2_2 240x320 x16 128->128", 128, 16 * 76800, 1152),
        ("conv3_2 120x160 x1  256->256", 256, 19200, 2304),
        ("conv3_2 120x160 x16 256->256", 256, 16 * 19200, 2304),
        ("conv4_2 60x80   x1  512->512", 512, 4800, 4608),
        ("conv4_2 60x80   x16 512->512", 512, 16 * 4800, 4608),
        ("conv5_2 30x40   x1  512->512", 512, 1200, 4608),
        ("conv5_2 30x40   x16 512->512", 512, 16 * 1200, 4608),
        ("conv1_2 480x640 x1  64->64", 64, 307200, 576),
    ):
        t, ms = tflops(M, N, K)
        print(f"{name:34s} M={M:5d} N={N:8d} K={K:5d}: {t:8.1f} TFLOP/s  {ms * 1e3:9.1f} us", flush=True)


if __name__ == "__main__":
    main()
