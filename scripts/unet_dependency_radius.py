"""Dependency radius of the UNet pyramid's outputs on the input image, in input pixels, by interval propagation through
the 13 encoder convolutions (3x3, pad 1), the four 2x2 max-pools, the four bilinear x2 up-samplings (align_corners =
False: output o reads sources (o - 1) // 2 .. (o + 1) // 2) and the four decoder convolutions over [up(prev) | skip]
(SURVEY A.5).  Maximum over the output pixel's phase within a 16-pixel period.  PoseTrackerRefiner.WINDOW_MARGIN must
cover the stride-1 figure + 1 (the bilinear sample's second texel) + 16 (window alignment).  CPU only."""
N_CONV = [2, 2, 3, 3, 3]


def conv(iv):
    return iv[0] - 1, iv[1] + 1


def pool_in(iv):
    return 2 * iv[0], 2 * iv[1] + 1


def up_in(iv):
    return (iv[0] - 1) // 2, (iv[1] + 1) // 2


def back_enc(iv, block):
    for b in range(block, -1, -1):
        for _ in range(N_CONV[b]):
            iv = conv(iv)
        if b > 0:
            iv = pool_in(iv)
    return iv


def back_dec(iv, d):
    iv = conv(iv)
    skip = back_enc(iv, 3 - d)
    prev = up_in(iv)
    p = back_enc(prev, 4) if d == 0 else back_dec(prev, d - 1)
    return min(skip[0], p[0]), max(skip[1], p[1])


def radii():
    out = {}
    for name, f, stride in (("stride 1 (dec3 + fine head)", lambda iv: back_dec(iv, 3), 1),
                            ("stride 4 (dec1 + mid head)", lambda iv: back_dec(iv, 1), 4),
                            ("stride 16 (enc4 + coarse head)", lambda iv: back_enc(iv, 4), 16)):
        worst = 0
        for o in range(4096, 4096 + 16):
            lo, hi = f((o, o))
            worst = max(worst, o * stride - lo, hi - (o * stride + stride - 1))
        out[name] = worst
    return out


if __name__ == "__main__":
    for k, v in radii().items():
        print(f"{k}: {v} input pixels")
