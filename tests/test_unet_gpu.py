"""Parity of the HIP UNet (MFMA implicit-GEMM convs, fp16 activations, fp32 accumulate)
against the fp32 PyTorch-CPU oracle.  Tolerances are those of fp16 storage (11-bit
mantissa, the same mantissa as the TF32 convs the reference runs on Ampere)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_oracle as UO
from pixtrack_amd import _lib
from pixtrack_amd.unet import OUTPUT_DIMS, UNet, make_synthetic_unet_weights

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W,Cin,Cout,relu", [(16, 16, 32, 32, 1), (37, 50, 64, 64, 1), (20, 33, 128, 96, 0),
                                               (30, 40, 512, 128, 1), (48, 64, 1024, 64, 1)])
def test_conv3x3_layer_matches_conv2d(device, H, W, Cin, Cout, relu):
    """KAT-6: one layer vs torch.nn.functional.conv2d on the same fp16-rounded operands;
    asymmetric random weights catch operand/row-column transposes."""
    g = torch.Generator().manual_seed(H * 1000 + W + Cin)
    x = torch.randn(Cin, H, W, generator=g).half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).half()
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.float()[None], w.float(), b, padding=1)[0]
    if relu:
        ref = F.relu(ref)
    xd = x.permute(1, 2, 0).contiguous().to(device)
    wd = w.permute(0, 2, 3, 1).contiguous().to(device)
    bd = b.to(device)
    out = torch.full((H, W, Cout), float("nan"), dtype=torch.float16, device=device)
    _lib.check(_lib.lib().pxt_conv3x3_nhwc_f16(xd.data_ptr(), H, W, Cin, wd.data_ptr(), bd.data_ptr(), Cout, relu,
                                               out.data_ptr(), _lib.stream_ptr(device)), "conv")
    torch.cuda.synchronize()
    got = out.float().cpu().permute(2, 0, 1)
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    assert err < 2e-3 * max(1.0, ref.abs().max().item()), err


def _compare_pyramid(device, H, W, normalize, seed=3, bn_trivial=False, mask=None, u8=False):
    w = make_synthetic_unet_weights(seed=seed, bn_trivial=bn_trivial)
    rng = np.random.default_rng(seed)
    img = rng.uniform(0, 255, size=(H, W, 3)).astype(np.float32)
    # smooth it a little so it resembles an image rather than white noise
    img = (img + np.roll(img, 1, 0) + np.roll(img, 1, 1) + np.roll(img, 2, 0)) / 4
    if u8:
        img = np.floor(img).astype(np.float32)
    img_ref = img * mask[..., None] if mask is not None else img
    feats, confs = UO.unet_forward(w, torch.from_numpy(img_ref).permute(2, 0, 1) / 255.0)
    net = UNet(w, device)
    src = torch.from_numpy(img.astype(np.uint8) if u8 else img).to(device)
    md = torch.from_numpy(mask.astype(np.uint8)).to(device) if mask is not None else None
    outs = net.forward_packed(src, md, normalize=normalize)
    torch.cuda.synchronize()
    for k, (o, c) in enumerate(zip(outs, OUTPUT_DIMS)):
        o = o.cpu()
        f_ref = feats[k].permute(1, 2, 0)
        assert o.shape[:2] == f_ref.shape[:2], (o.shape, f_ref.shape)
        f = o[..., :c]
        if normalize:
            f_ref = f_ref / f_ref.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        cos = F.cosine_similarity(f, f_ref, dim=-1)
        assert cos.min().item() > 0.9995, (k, cos.min().item())
        scale = f_ref.abs().max().item()
        assert (f - f_ref).abs().max().item() < 2e-2 * scale, (k, (f - f_ref).abs().max().item(), scale)
        assert (o[..., c] - confs[k][0]).abs().max().item() < 5e-3
        assert (o[..., c + 1:] == 0).all()
    return outs


@pytest.mark.parametrize("H,W", [(64, 48), (96, 128)])
def test_unet_matches_oracle(device, H, W):
    _compare_pyramid(device, H, W, normalize=False)


def test_unet_odd_size_crops_like_pixloc(device):
    """Sizes not divisible by 16: pool floors, decoder crops the skip (A.5)."""
    _compare_pyramid(device, 75, 100, normalize=True)


def test_unet_mask_and_u8_and_normalize(device):
    rng = np.random.default_rng(0)
    mask = (rng.uniform(size=(64, 80)) > 0.3).astype(np.float32)
    _compare_pyramid(device, 64, 80, normalize=True, mask=mask, u8=True)


def test_unet_module_call_convention(device):
    """model({"image": 1x3xHxW in [0,1]}) -> feature_maps / confidences, model.scales."""
    w = make_synthetic_unet_weights(seed=5)
    net = UNet(w, device)
    x = torch.rand(1, 3, 48, 64)
    pred = net({"image": x.to(device)})
    feats, confs = UO.unet_forward(w, x[0])
    assert net.scales == [1, 4, 16]
    for k in range(3):
        assert pred["feature_maps"][k].shape == (1,) + tuple(feats[k].shape)
        assert pred["confidences"][k].shape == (1,) + tuple(confs[k].shape)
        cos = F.cosine_similarity(pred["feature_maps"][k][0].cpu(), feats[k], dim=0)
        assert cos.min().item() > 0.9995


def test_unet_config5_resolution(device):
    """1024x576 (what a 1920x1080 query becomes after the extractor's resize)."""
    _compare_pyramid(device, 576, 1024, normalize=True, seed=9, bn_trivial=True)


@pytest.mark.parametrize("H,W", [(96, 128), (48, 80)])
def test_unet_batch_matches_single_images_and_oracle(device, H, W):
    """pxt_unet_forward_batch: a u8 image and a masked float image in one pass.  Each image's
    maps equal its single-image maps up to fp32 summation order (the split-K factor depends
    on the batch), do not depend on the batch neighbour, and match the CPU oracle."""
    w = make_synthetic_unet_weights(seed=5)
    rng = np.random.default_rng(9)
    a = np.floor(rng.uniform(0, 255, size=(H, W, 3))).astype(np.uint8)
    b = rng.uniform(0, 255, size=(H, W, 3)).astype(np.float32)
    mask = (rng.uniform(size=(H, W)) > 0.3).astype(np.uint8)
    net = UNet(w, device)
    ta, tb, tm = torch.from_numpy(a).to(device), torch.from_numpy(b).to(device), torch.from_numpy(mask).to(device)
    single_a = [o.clone() for o in net.forward_packed(ta, None, normalize=False)]
    single_b = [o.clone() for o in net.forward_packed(tb, tm, normalize=True)]
    both = net.forward_packed_batch([(ta, None, False), (tb, tm, True)])
    swapped = net.forward_packed_batch([(tb, tm, True), (ta, None, False)])
    other = net.forward_packed_batch([(ta, None, False), (ta, None, False)])
    torch.cuda.synchronize()
    for k in range(3):
        for got, ref in ((both[0][k], single_a[k]), (both[1][k], single_b[k])):
            scale = ref.abs().max().item()
            assert (got - ref).abs().max().item() < 2e-3 * scale, k
        assert torch.equal(both[0][k], swapped[1][k]) and torch.equal(both[1][k], swapped[0][k])
        assert torch.equal(both[0][k], other[0][k]) and torch.equal(other[0][k], other[1][k])
    feats, confs = UO.unet_forward(w, torch.from_numpy(b * mask[..., None]).permute(2, 0, 1) / 255.0)
    for k, c in enumerate(OUTPUT_DIMS):
        f_ref = feats[k].permute(1, 2, 0)
        f_ref = f_ref / f_ref.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        cos = F.cosine_similarity(both[1][k][..., :c].cpu(), f_ref, dim=-1)
        assert cos.min().item() > 0.9995, (k, cos.min().item())
        assert (both[1][k][..., c].cpu() - confs[k][0]).abs().max().item() < 5e-3


def test_unet_pair_of_two_sizes_matches_single_images(device):
    """pxt_unet_forward_pair: a frame's reference render and its masked query have different sizes with real assets
    (reference camera x 0.5 / x 0.3); the two single-image passes then run side by side on two streams.  Each image's
    maps equal its single-image maps up to fp32 summation order, in either order of the pair, run after run."""
    w = make_synthetic_unet_weights(seed=5)
    rng = np.random.default_rng(11)
    a = np.floor(rng.uniform(0, 255, size=(144, 191, 3))).astype(np.uint8)   # odd width: cropped like pixloc
    b = rng.uniform(0, 255, size=(240, 320, 3)).astype(np.float32)
    mask = (rng.uniform(size=(240, 320)) > 0.3).astype(np.uint8)
    net = UNet(w, device)
    ta, tb, tm = torch.from_numpy(a).to(device), torch.from_numpy(b).to(device), torch.from_numpy(mask).to(device)
    single_a = [o.clone() for o in net.forward_packed(ta, None, normalize=False)]
    single_b = [o.clone() for o in net.forward_packed(tb, tm, normalize=True)]
    both = net.forward_packed_batch([(ta, None, False), (tb, tm, True)])
    swapped = net.forward_packed_batch([(tb, tm, True), (ta, None, False)])
    again = net.forward_packed_batch([(ta, None, False), (tb, tm, True)])
    torch.cuda.synchronize()
    for k in range(3):
        assert both[0][k].shape == single_a[k].shape and both[1][k].shape == single_b[k].shape
        for got, ref in ((both[0][k], single_a[k]), (both[1][k], single_b[k])):
            scale = ref.abs().max().item()
            assert (got - ref).abs().max().item() < 2e-3 * scale, k
        assert torch.equal(both[0][k], swapped[1][k]) and torch.equal(both[1][k], swapped[0][k])
        assert torch.equal(both[0][k], again[0][k]) and torch.equal(both[1][k], again[1][k])


def _packed_conv(device, x, w, b, relu, cfg=0, splits=1, pool=False):
    """x [Cin,H,W] fp16, w [Cout,Cin,3,3] fp16 -> (out [Cout,H,W] float, pooled or None) through
    pxt_conv3x3_pack_weights + pxt_conv3x3_packed with an explicit tile configuration."""
    L = _lib.lib()
    Cin, H, W = x.shape
    Cout = w.shape[0]
    xd = x.permute(1, 2, 0).contiguous().to(device)
    wd = w.permute(0, 2, 3, 1).contiguous().to(device)
    bd = b.to(device)
    packed = torch.empty(int(L.pxt_conv3x3_packed_bytes(Cin, Cout)), dtype=torch.uint8, device=device)
    _lib.check(L.pxt_conv3x3_pack_weights(wd.data_ptr(), Cin, Cout, packed.data_ptr(), _lib.stream_ptr(device)), "pack")
    out = torch.full((H, W, Cout), float("nan"), dtype=torch.float16, device=device)
    pl = torch.full((H // 2, W // 2, Cout), float("nan"), dtype=torch.float16, device=device) if pool else None
    ws = torch.empty(max(1, splits) * H * W * Cout * 4, dtype=torch.uint8, device=device) if splits > 1 else None
    _lib.check(L.pxt_conv3x3_packed(xd.data_ptr(), H, W, Cin, packed.data_ptr(), bd.data_ptr(), Cout, relu, out.data_ptr(),
                                    _lib.dptr(pl), cfg, splits, _lib.dptr(ws), ws.numel() if ws is not None else 0,
                                    _lib.stream_ptr(device)), "conv packed")
    torch.cuda.synchronize()
    return out.float().cpu().permute(2, 0, 1), (pl.float().cpu().permute(2, 0, 1) if pool else None)


@pytest.mark.parametrize("cfg", [1, 2, 4, 6, 11, 13, 14, 15, 16, 17, 18, 19])
@pytest.mark.parametrize("H,W,Cin,Cout", [(37, 50, 64, 128), (64, 48, 96, 256), (9, 130, 32, 128)])
def test_conv3x3_every_tile_configuration(device, cfg, H, W, Cin, Cout):
    """Every workgroup tiling of the second kernel (1-6) and of the third (11-16: shared pixel fragments, filter
    fragments through LDS, 16- or 32-channel chunks) gives the same layer, on ragged sizes (partial tiles in both
    directions), with and without the fused 2x2 max-pool and with split-K."""
    g = torch.Generator().manual_seed(cfg * 7919 + H * 1000 + W + Cin)
    x = torch.randn(Cin, H, W, generator=g).half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).half()
    b = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x.float()[None], w.float(), b, padding=1)[0])
    tol = 2e-3 * max(1.0, ref.abs().max().item())
    got, pooled = _packed_conv(device, x, w, b, 1, cfg=cfg, pool=True)
    assert torch.isfinite(got).all() and (got - ref).abs().max().item() < tol
    # the pooled copy is the max-pool of the fp16 output itself: exact
    want_pool = F.max_pool2d(got[None], 2)[0]
    if cfg not in (15, 17, 18):  # (odd number of row blocks per wave: no fused pool, the pyramid pools separately)
        assert torch.isfinite(pooled).all() and torch.equal(pooled, want_pool)
    # (18 / 19 split the K range over two wave quartets as well: every split needs an even number of 32-channel chunks)
    splits = min(3, Cin // 32) if cfg < 18 else 1
    got2, _ = _packed_conv(device, x, w, b, 1, cfg=cfg, splits=splits)
    assert (got2 - ref).abs().max().item() < tol
    # without ReLU / without pool: same numbers as the pooled run where positive
    got3, _ = _packed_conv(device, x, w, b, 0, cfg=cfg)
    ref3 = F.conv2d(x.float()[None], w.float(), b, padding=1)[0]
    assert (got3 - ref3).abs().max().item() < tol


@pytest.mark.parametrize("cfg,splits", [(18, 2), (18, 4), (19, 2), (15, 4), (11, 8)])
def test_split_k_on_the_third_kernel(device, cfg, splits):
    """The 30x40 layers' shape: split-K on the third kernel - also with the K range of every split divided between two wave
    quartets (18 / 19) - summed by the reduction kernel in split order: bit-identical run to run."""
    H, W, Cin, Cout = 30, 40, 512, 256
    g = torch.Generator().manual_seed(cfg * 31 + splits)
    x = torch.randn(Cin, H, W, generator=g).half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).half()
    b = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x.float()[None], w.float(), b, padding=1)[0])
    tol = 2e-3 * max(1.0, ref.abs().max().item())
    first = None
    for _ in range(3):
        got, _ = _packed_conv(device, x, w, b, 1, cfg=cfg, splits=splits)
        assert torch.isfinite(got).all() and (got - ref).abs().max().item() < tol
        first = got if first is None else first
        assert torch.equal(got, first)


def test_activation_stats_flag_an_fp16_overflow(device):
    """UNet.activation_stats (pxt_unet_activation_stats): the range check a maintainer runs on a NEW checkpoint - pixloc
    runs this network in fp32, the library stores activations as fp16.  Synthetic weights stay far inside the range; the
    same weights with every 3x3 filter scaled by 12 overflow in the deeper layers, and the check says where."""
    from pixtrack_amd.unet import UNet, make_synthetic_unet_weights

    w = make_synthetic_unet_weights(7)
    img = (torch.rand(120, 160, 3, generator=torch.Generator().manual_seed(3)) * 255).to(device)
    st = UNet(w, device).activation_stats(img)
    assert len(st) == 17 and st[0] is None and st[16] is None  # fused first layer / fused fine head: never in memory
    seen = [s for s in st if s is not None]
    assert len(seen) == 15 and all(n == 0 for _, n in seen) and all(1e-3 < m < 6.0e4 for m, _ in seen), st
    big = {k: (v * 12.0 if (v.dim() == 4 and v.shape[-1] == 3) else v) for k, v in w.items()}
    st2 = UNet(big, device).activation_stats(img)
    bad = [l for l, s in enumerate(st2) if s is not None and s[1] > 0]
    assert bad and min(bad) >= 2, st2  # the first layers still fit, the growth of 12x per layer overflows further down


def test_auto_rescale_brings_an_overflowing_checkpoint_into_fp16_range(device):
    """A checkpoint whose activations overflow fp16 storage (every 3x3 filter of the synthetic weights x 3.5: the deeper
    layers reach 1e6 in fp32) gives non-finite maps as it is; after auto_rescale_for_fp16 - exact power-of-two storage
    factors found with activation_stats - the HIP pyramid matches the fp32 oracle of the ORIGINAL overflowing weights."""
    from pixtrack_amd.unet import auto_rescale_for_fp16

    w = make_synthetic_unet_weights(7)
    big = {k: (v * 3.5 if (v.dim() == 4 and v.shape[-1] == 3) else v) for k, v in w.items()}
    rng = np.random.default_rng(5)
    img = rng.uniform(0, 255, size=(96, 128, 3)).astype(np.float32)
    img = (img + np.roll(img, 1, 0) + np.roll(img, 1, 1) + np.roll(img, 2, 0)) / 4
    src = torch.from_numpy(img).to(device)
    raw = UNet(big, device).forward_packed(src, None, normalize=True)
    assert not all(bool(torch.isfinite(o).all()) for o in raw)
    fixed, c = auto_rescale_for_fp16(big, device, [src])
    assert max(c) >= 256.0 and all(math.log2(x) == int(math.log2(x)) for x in c)
    outs = UNet(fixed, device).forward_packed(src, None, normalize=True)
    feats, confs = UO.unet_forward(big, torch.from_numpy(img).permute(2, 0, 1) / 255.0)
    for k, (o, ch) in enumerate(zip(outs, OUTPUT_DIMS)):
        o = o.cpu()
        assert torch.isfinite(o).all()
        f_ref = feats[k].permute(1, 2, 0)
        f_ref = f_ref / f_ref.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        cos = F.cosine_similarity(o[..., :ch], f_ref, dim=-1)
        assert cos.min().item() > 0.9995, (k, cos.min().item())
        # the confidences of this artificial network are saturated (logits of +-1e5: sigmoid is 0 or 1, and fp16 noise
        # flips the few pixels whose logit happens to be near zero): equal on all but a sliver of the map
        assert ((o[..., ch] - confs[k][0]).abs() > 5e-3).float().mean().item() < 0.01


def test_deferred_join_of_the_pair_pass(device):
    """pxt_unet_set_defer_join / pxt_unet_pair_join: the pair call returns with the first image's maps complete in stream
    order and the second pass still running; after join() both equal what the joined call produces, bit for bit - and
    the extractor's staged path (reference first, announced query second) uses it and joins when the query's maps are
    asked for."""
    from pixtrack_amd.feature_extractor import PixTrackFeatureExtractor

    net = UNet(make_synthetic_unet_weights(7), device)
    g = torch.Generator().manual_seed(9)
    a = (torch.rand(240, 320, 3, generator=g) * 255).to(torch.uint8).to(device)
    b = (torch.rand(192, 256, 3, generator=g) * 255).to(device)
    m = (torch.rand(192, 256, generator=g) > 0.3).to(torch.uint8).to(device)
    want = net.forward_packed_batch([(a, None, False), (b, m, True)])
    want = [[t.clone() for t in per] for per in want]
    net.set_defer_join(True)
    got = net.forward_packed_batch([(a, None, False), (b, m, True)])
    net.set_defer_join(False)
    first = [t.clone() for t in got[0]]  # (current-stream work on image 0's maps: legal before the join)
    net.join()
    torch.cuda.synchronize()
    for x, y in zip(first + got[1], want[0] + want[1]):
        assert torch.equal(x, y)
    net.join()  # nothing pending: a no-op
    ex = PixTrackFeatureExtractor(net, device)
    assert ex.defer_join
    ex.stage(b, 1, m, True)
    ref_maps, _ = ex.extract_packed(a, 1, None, False)
    assert ex._join_pending
    q_maps, _ = ex.extract_packed(b, 1, m, True)
    assert not ex._join_pending
    torch.cuda.synchronize()
    for x, y in zip(list(ref_maps) + list(q_maps), want[0] + want[1]):
        assert torch.equal(x, y)


def _tracker_like_images(H, W, seed, device):
    """A 'reference render' (uint8, exactly 0 outside a blob) and a 'masked query' (float noise + a dilated silhouette
    mask), as the tracker hands them to the extractor (pixloc_tracker_r9.py:224-227)."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    cy, cx, ry, rx = 0.52 * H, 0.47 * W, 0.17 * H, 0.2 * W
    blob = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) < 1.0
    ref = (torch.rand(H, W, 3, generator=g) * 255).to(torch.uint8) * blob[..., None].to(torch.uint8)
    mask = (((yy - cy) / (ry + 12)) ** 2 + ((xx - cx) / (rx + 12)) ** 2) < 1.0
    query = torch.rand(H, W, 3, generator=g) * 255
    return ref.to(device).contiguous(), query.to(device).contiguous(), mask.to(torch.uint8).to(device).contiguous()


@pytest.mark.parametrize("H,W", [(480, 640), (240, 320), (333, 421)])
def test_constant_tile_skipping_leaves_every_map_bit_identical(device, H, W):
    """Encoder blocks 1-3 skip tiles whose dependency cone lies where the input is constant (pxt_unet_set_tile_skip):
    single passes, the frame's two-stream pair pass and a batched pass, skip on against skip off - torch.equal."""
    net = UNet(make_synthetic_unet_weights(7), device)
    ref, query, mask = _tracker_like_images(H, W, 3, device)
    calls = {
        "single u8": lambda: [net.forward_packed(ref, None, False)],
        "single masked": lambda: [net.forward_packed(query, mask, True)],
        "pair": lambda: net.forward_packed_batch([(ref, None, False), (query, mask, True)]),
        "batch of 4": lambda: net.forward_packed_batch([(ref, None, False), (query, mask, True), (query, None, True),
                                                       (ref, None, True)]),
    }
    for name, call in calls.items():
        net.set_tile_skip(True)
        on = [[m.clone() for m in maps] for maps in call()]
        net.set_tile_skip(False)
        off = call()
        torch.cuda.synchronize()
        for a, b in zip(on, off):
            for l, (x, y) in enumerate(zip(a, b)):
                assert torch.equal(x, y), (name, l, float((x - y).abs().max()))
    net.set_tile_skip(True)


def test_constant_tile_skipping_two_sizes_and_a_fully_masked_image(device):
    """The pair entry with two image sizes (real assets), and the degenerate inputs: a mask that is 0 everywhere (every
    interior tile constant) and one that is 1 everywhere (nothing to skip)."""
    net = UNet(make_synthetic_unet_weights(7), device)
    ref, _q, _m = _tracker_like_images(496, 512, 5, device)
    _r, query, mask = _tracker_like_images(480, 640, 6, device)
    cases = [[(ref, None, False), (query, mask, True)],
             [(query, torch.zeros_like(mask), True), (query, torch.ones_like(mask), True)]]
    for items in cases:
        net.set_tile_skip(True)
        on = [[m.clone() for m in maps] for maps in net.forward_packed_batch(items)]
        net.set_tile_skip(False)
        off = net.forward_packed_batch(items)
        for a, b in zip(on, off):
            for x, y in zip(a, b):
                assert torch.equal(x, y)
    net.set_tile_skip(True)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_constant_tile_skipping_isolated_pixels_probe_the_cone_radii(device, seed):
    """Adversarial activity: a handful of isolated non-zero pixels / tiny mask rectangles at random places (next to tile
    corners as often as not).  A dependency radius one pixel short anywhere in the six layers would skip a tile a lit
    pixel reaches, and the maps would differ."""
    H, W = 368, 496
    g = torch.Generator().manual_seed(100 + seed)
    net = UNet(make_synthetic_unet_weights(7), device)
    ref = torch.zeros(H, W, 3, dtype=torch.uint8)
    for _ in range(10):
        y, x = int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, W, (1,), generator=g))
        ref[y, x, int(torch.randint(0, 3, (1,), generator=g))] = int(torch.randint(1, 256, (1,), generator=g))
    mask = torch.zeros(H, W, dtype=torch.uint8)
    for _ in range(6):
        y, x = int(torch.randint(0, H - 3, (1,), generator=g)), int(torch.randint(0, W - 3, (1,), generator=g))
        mask[y:y + int(torch.randint(1, 4, (1,), generator=g)), x:x + int(torch.randint(1, 4, (1,), generator=g))] = 1
    query = torch.rand(H, W, 3, generator=g) * 255
    items = [(ref.to(device), None, False), (query.to(device), mask.to(device), True)]
    net.set_tile_skip(True)
    on = [[m.clone() for m in maps] for maps in net.forward_packed_batch(items)]
    single_on = [m.clone() for m in net.forward_packed(items[0][0], None, False)]
    net.set_tile_skip(False)
    off = net.forward_packed_batch(items)
    single_off = net.forward_packed(items[0][0], None, False)
    for a, b in zip(on + [single_on], off + [single_off]):
        for x, y in zip(a, b):
            assert torch.equal(x, y), float((x - y).abs().max())
    net.set_tile_skip(True)


def test_resized_images_keep_their_constant_regions_for_the_tile_skipping(device):
    """Images above the extractor's limit are resized (feature_extractor.py:41-45) and lose mask and uint8 type; the
    extractor hands the UNet a conservative 'may be non-zero' plane instead (pxt_resize_activity).  It must cover every
    non-zero pixel of the resized image, leave most of a tracker-like frame inactive, and the maps must not depend on it."""
    from pixtrack_amd.feature_extractor import PixTrackFeatureExtractor

    net = UNet(make_synthetic_unet_weights(7), device)
    ex = PixTrackFeatureExtractor(net, device)
    ref, query, mask = _tracker_like_images(1080, 1920, 9, device)
    for image, m in ((query, mask), (ref, None)):
        img, act, _sr = ex._prepare(image, 1, m)
        assert tuple(img.shape[:2]) == (576, 1024) and act is not None and act.dtype == torch.uint8
        nonzero = (img != 0).any(-1)
        assert not bool((nonzero & (act == 0)).any())        # every pixel that is not exactly 0 is marked active
        assert 0.02 < float(act.float().mean()) < 0.5          # ... and most of the frame is not
        net.set_tile_skip(True)
        on = [x.clone() for x in net.forward_packed(img, act, True)]
        net.set_tile_skip(False)
        off = net.forward_packed(img, act, True)
        plain = net.forward_packed(img, None, True)            # (the plane multiplies by exactly 0 or 1: same maps without it)
        for a, b, c in zip(on, off, plain):
            assert torch.equal(a, b) and torch.equal(a, c)
    net.set_tile_skip(True)
