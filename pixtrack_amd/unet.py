"""Host side of the UNet feature pyramid (pixloc ``UNet``, experiment pixloc_megadepth).

Reference interface: ``pred = self.model({"image": image_tensor})`` ->
``{"feature_maps": [...], "confidences": [...]}`` with ``model.scales``
(pixtrack/localization/feature_extractor.py:26,48-57).  Architecture: SURVEY.md A.5.

The forward pass is a chain of hand-written HIP kernels behind ``pxt_unet_forward``
(csrc/pxt_unet.hip): fp16 NHWC activations, MFMA implicit-GEMM 3x3 convolutions with
fp32 accumulation, BatchNorm folded into the decoder convolutions, and 1x1 heads that
write the HWC float32 maps (descriptor + confidence channel) the LM kernel reads.
This module only packs weights and marshals pointers; there is no PyTorch conv path.
"""
from __future__ import annotations

import ctypes as C
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .ops import ops
from .optimizer import cstride_for

ENC_BLOCKS = [[(3, 64), (64, 64)], [(64, 128), (128, 128)], [(128, 256), (256, 256), (256, 256)],
              [(256, 512), (512, 512), (512, 512)], [(512, 512), (512, 512), (512, 512)]]
SKIP_DIMS = [64, 128, 256, 512, 512]
DECODER = [64, 64, 64, 32]
OUTPUT_SCALES = [0, 2, 4]
OUTPUT_DIMS = [32, 128, 128]
HEAD_INPUTS = [32, 64, 512]
BN_EPS = 1e-5
MAGIC = b"PXTUNET1"


def conv_layer_names() -> List[str]:
    names = [f"enc{b}_{i}" for b, convs in enumerate(ENC_BLOCKS) for i in range(len(convs))]
    return names + [f"dec{d}" for d in range(len(DECODER))]


def conv_layer_dims() -> List[Tuple[int, int]]:
    dims = [c for convs in ENC_BLOCKS for c in convs]
    prev = SKIP_DIMS[-1]
    for out, skip in zip(DECODER, SKIP_DIMS[:-1][::-1]):
        dims.append((prev + skip, out))
        prev = out
    return dims


def make_synthetic_unet_weights(seed: int = 7, bn_trivial: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded stand-in for the absent pixloc_megadepth checkpoint: He-normal
    convolutions, zero biases for the encoder... BN gamma=1, beta=0, running stats 0/1
    (``bn_trivial=False`` randomises the BN statistics to exercise the folding)."""
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, torch.Tensor] = {}

    def he(cout, cin, k):
        return torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5

    for b, convs in enumerate(ENC_BLOCKS):
        for i, (cin, cout) in enumerate(convs):
            w[f"enc{b}_{i}.weight"] = he(cout, cin, 3)
            w[f"enc{b}_{i}.bias"] = 0.05 * torch.randn(cout, generator=g)
    prev = SKIP_DIMS[-1]
    for d, (out, skip) in enumerate(zip(DECODER, SKIP_DIMS[:-1][::-1])):
        w[f"dec{d}.weight"] = he(out, prev + skip, 3)
        if bn_trivial:
            w[f"dec{d}.bn_weight"] = torch.ones(out)
            w[f"dec{d}.bn_bias"] = torch.zeros(out)
            w[f"dec{d}.bn_mean"] = torch.zeros(out)
            w[f"dec{d}.bn_var"] = torch.ones(out)
        else:
            w[f"dec{d}.bn_weight"] = 1.0 + 0.2 * torch.randn(out, generator=g)
            w[f"dec{d}.bn_bias"] = 0.1 * torch.randn(out, generator=g)
            w[f"dec{d}.bn_mean"] = 0.1 * torch.randn(out, generator=g)
            w[f"dec{d}.bn_var"] = 0.5 + torch.rand(out, generator=g)
        prev = out
    for k, (cin, dim) in enumerate(zip(HEAD_INPUTS, OUTPUT_DIMS)):
        w[f"adapt{k}.weight"] = torch.randn(dim, cin, 1, 1, generator=g) * (1.0 / cin) ** 0.5
        w[f"adapt{k}.bias"] = 0.05 * torch.randn(dim, generator=g)
        w[f"unc{k}.weight"] = torch.randn(1, cin, 1, 1, generator=g) * (1.0 / cin) ** 0.5
        w[f"unc{k}.bias"] = 0.05 * torch.randn(1, generator=g)
    return w


def from_pixloc_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Maps a pixloc checkpoint's ``extractor.*`` keys (torchvision VGG16 indices inside
    checkpointed Sequential blocks) to the canonical names used here."""
    sd = {k[len("extractor."):] if k.startswith("extractor.") else k: v for k, v in sd.items()}
    out: Dict[str, torch.Tensor] = {}
    for b, convs in enumerate(ENC_BLOCKS):
        # block 0: conv,relu,conv,relu -> conv indices 0,2 ; later blocks open with the pool:
        # pool,conv,relu,... -> conv indices 1,3,5
        first = 0 if b == 0 else 1
        for i in range(len(convs)):
            idx = first + 2 * i
            out[f"enc{b}_{i}.weight"] = sd[f"encoder.{b}.{idx}.weight"]
            out[f"enc{b}_{i}.bias"] = sd[f"encoder.{b}.{idx}.bias"]
    for d in range(len(DECODER)):
        out[f"dec{d}.weight"] = sd[f"decoder.{d}.layers.0.weight"]
        out[f"dec{d}.bn_weight"] = sd[f"decoder.{d}.layers.1.weight"]
        out[f"dec{d}.bn_bias"] = sd[f"decoder.{d}.layers.1.bias"]
        out[f"dec{d}.bn_mean"] = sd[f"decoder.{d}.layers.1.running_mean"]
        out[f"dec{d}.bn_var"] = sd[f"decoder.{d}.layers.1.running_var"]
    for k in range(len(OUTPUT_SCALES)):
        out[f"adapt{k}.weight"] = sd[f"adaptation.{k}.0.weight"]
        out[f"adapt{k}.bias"] = sd[f"adaptation.{k}.0.bias"]
        out[f"unc{k}.weight"] = sd[f"uncertainty.{k}.0.weight"]
        out[f"unc{k}.bias"] = sd[f"uncertainty.{k}.0.bias"]
    return out


def to_pixloc_state_dict(w: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Inverse of from_pixloc_state_dict (``extractor.*`` + ``optimizer.*`` keys): what a pixloc
    ``checkpoint_best.tar`` holds under "model".  Used to write test checkpoints."""
    out: Dict[str, torch.Tensor] = {}
    for b, convs in enumerate(ENC_BLOCKS):
        first = 0 if b == 0 else 1
        for i in range(len(convs)):
            for part in ("weight", "bias"):
                out[f"extractor.encoder.{b}.{first + 2 * i}.{part}"] = w[f"enc{b}_{i}.{part}"]
    for d in range(len(DECODER)):
        out[f"extractor.decoder.{d}.layers.0.weight"] = w[f"dec{d}.weight"]
        for src, dst in (("bn_weight", "weight"), ("bn_bias", "bias"), ("bn_mean", "running_mean"),
                         ("bn_var", "running_var")):
            out[f"extractor.decoder.{d}.layers.1.{dst}"] = w[f"dec{d}.{src}"]
        out[f"extractor.decoder.{d}.layers.1.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    for k in range(len(OUTPUT_SCALES)):
        for part in ("weight", "bias"):
            out[f"extractor.adaptation.{k}.0.{part}"] = w[f"adapt{k}.{part}"]
            out[f"extractor.uncertainty.{k}.0.{part}"] = w[f"unc{k}.{part}"]
    for k, v in w.items():
        if k.startswith("optimizer."):
            out[k] = v
    return out


def load_weights(path) -> Dict[str, torch.Tensor]:
    """Reads either this package's flat tensor dict or a pixloc experiment checkpoint
    (``checkpoint_best.tar``: {"model": state_dict, "conf": ...}; reference
    pixloc_pose_refiners.py:49-60 loads it through pixloc's load_experiment).  Returns the
    canonical names of this module plus any ``optimizer.{i}.dampingnet.const``."""
    blob = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(blob, dict) and "model" in blob and isinstance(blob["model"], dict):
        blob = blob["model"]
    if any(k.startswith("extractor.") for k in blob):
        out = from_pixloc_state_dict(blob)
        out.update({k: v for k, v in blob.items() if k.startswith("optimizer.")})
        return out
    return blob


# ---------------------------------------------------------------------------------------------
# fp16 range of a NEW checkpoint.  pixloc runs this network in fp32; the kernels here store activations as fp16
# (max 65504).  A ReLU network is positively homogeneous layer by layer - ReLU, max-pool, bilinear up-sampling and
# concatenation commute with a positive scale - so the activations of layer l can be stored divided by any c_l > 0 if
# the layer's filters and bias are scaled by (scale of its input) / c_l and every consumer multiplies its input channels
# back.  With powers of two that is EXACT in fp32 and in the fp16 weights: the network computes the same function, its
# intermediate values sit where fp16 can hold them.  `auto_rescale_for_fp16` finds the c_l with UNet.activation_stats.
_BLOCK_LAST = [1, 3, 6, 9, 12]  # index of each encoder block's last convolution (the skip / head sources)


def rescale_unet_weights(w: Dict[str, torch.Tensor], c: List[float]) -> Dict[str, torch.Tensor]:
    """The same network with layer l's activations stored divided by c[l] (17 positive factors, powers of two for
    exactness; c = 1 everywhere returns equal tensors).  Heads absorb the factor of their source layer, so the
    outputs are unchanged."""
    names = conv_layer_names()
    assert len(c) == len(names) == 17 and all(x > 0 for x in c)
    out = dict(w)
    for l in range(13):  # encoder: input = the previous layer's output (pooled or not), the image for layer 0
        c_in = 1.0 if l == 0 else c[l - 1]
        out[f"{names[l]}.weight"] = w[f"{names[l]}.weight"] * (c_in / c[l])
        out[f"{names[l]}.bias"] = w[f"{names[l]}.bias"] / c[l]
    prev_l, prev_ch = 12, SKIP_DIMS[-1]
    for d in range(len(DECODER)):  # decoder d: input = concat[upsampled previous (c_prev), skip of block 3 - d (c_skip)]
        l = 13 + d
        skip_l = _BLOCK_LAST[3 - d]
        W = w[f"dec{d}.weight"].clone()
        W[:, :prev_ch] *= c[prev_l]
        W[:, prev_ch:] *= c[skip_l]
        out[f"dec{d}.weight"] = W
        # BatchNorm after the convolution: y = (conv - mean) * gamma / sigma + beta; y / c = gamma / c, beta / c
        out[f"dec{d}.bn_weight"] = w[f"dec{d}.bn_weight"] / c[l]
        out[f"dec{d}.bn_bias"] = w[f"dec{d}.bn_bias"] / c[l]
        prev_l, prev_ch = l, DECODER[d]
    for k, src in enumerate((16, 14, 12)):  # heads: fine <- dec3, mid <- dec1, coarse <- enc4
        out[f"adapt{k}.weight"] = w[f"adapt{k}.weight"] * c[src]
        out[f"unc{k}.weight"] = w[f"unc{k}.weight"] * c[src]
    return out


def auto_rescale_for_fp16(w: Dict[str, torch.Tensor], device, images, target: float = 64.0, limit: float = 8192.0,
                          max_rounds: int = 64):
    """Power-of-two storage factors c[l] under which every observed layer of every calibration image (HWC 0..255
    device tensors) stays finite and below `limit`, found by repeated UNet.activation_stats passes: the first layer that
    overflows or exceeds the limit is scaled down (to ~`target` when its maximum is finite, by 256 when it is not) and
    the network rebuilt.  Returns (rescaled weights, c).  The two layers that are never written to memory are not
    observed: the first one keeps c = 1 (its input is the normalised image), the last decoder layer is scaled when the
    outputs are non-finite although every observed layer is in range."""
    import math

    c = [1.0] * 17
    for _ in range(max_rounds):
        net = UNet(rescale_unet_weights(w, c), device)
        worst = None
        for img in images:
            for l, s in enumerate(net.activation_stats(img)):
                if s is None:
                    continue
                mx, bad = s
                if bad > 0 or mx > limit:
                    if worst is None or l < worst[0]:
                        worst = (l, mx, bad)
        if worst is None:
            outs = [net.forward_packed(img, None, False) for img in images]
            if not all(bool(torch.isfinite(o).all()) for per in outs for o in per):
                # every observed layer is in range: it is the last decoder layer, whose fp16 output only ever exists in
                # the fused fine head's registers - it follows its input's factor, then goes up in steps of 16
                c[16] = max(c[16] * 16.0, c[15])
                if c[16] > 2.0 ** 40:
                    raise _lib.PxtError("auto_rescale_for_fp16: outputs stay non-finite although every observed layer is "
                                        "in range")
                continue
            fixed = rescale_unet_weights(w, c)
            too_big = [k for k, v in fixed.items() if k.endswith(".weight") and v.dim() == 4 and float(v.abs().max()) > 6.0e4]
            if too_big:  # (the compensating factors live in fp16 weights: a cumulative factor beyond ~2^17 does not fit)
                raise _lib.PxtError(f"auto_rescale_for_fp16: the compensated weights of {too_big} leave fp16's range "
                                    f"(factors {c}); this checkpoint needs wider storage than fp16")
            return fixed, c
        l, mx, bad = worst
        f = 256.0 if bad > 0 else 2.0 ** math.ceil(math.log2(mx / target))
        for j in range(l, 17):  # every later layer follows (its compensated weights then stay what they were: a network
            c[j] *= f           # whose magnitudes grow keeps growing, and no compensating factor piles up in fp16 weights)
    raise _lib.PxtError(f"auto_rescale_for_fp16: no stable scaling after {max_rounds} rounds (factors {c})")


def _align16(n: int) -> int:
    return (n + 15) // 16 * 16


def pack_unet_weights(w: Dict[str, torch.Tensor]) -> bytes:
    """Flat blob consumed by ``pxt_unet_create``.

    Layout: MAGIC(8) | int32 n_conv(17) | int32 n_heads(3) | n_conv x (int32 cin, cout) |
    n_heads x (int32 cin, cout) | n_arrays x (int64 offset, int64 bytes) | data.
    Arrays, in order: for every conv layer [weights, bias]; for every head [weights, bias].
      * layer 0 weights: float32 [cout][ky][kx][cin]
      * other conv weights: float16 [cout][ky][kx][cin]  (BatchNorm folded for decoders)
      * conv bias: float32 [cout]
      * head weights: float32 [cin][cout+1] (column `cout` = uncertainty row); bias [cout+1]
    """
    names, dims = conv_layer_names(), conv_layer_dims()
    arrays: List[bytes] = []
    for li, (name, (cin, cout)) in enumerate(zip(names, dims)):
        W = w[f"{name}.weight"].detach().float()
        assert tuple(W.shape) == (cout, cin, 3, 3), (name, W.shape)
        if name.startswith("dec"):
            s = w[f"{name}.bn_weight"].float() / torch.sqrt(w[f"{name}.bn_var"].float() + BN_EPS)
            b = w[f"{name}.bn_bias"].float() - w[f"{name}.bn_mean"].float() * s
            W = W * s[:, None, None, None]
        else:
            b = w[f"{name}.bias"].detach().float()
        Wk = W.permute(0, 2, 3, 1).contiguous()  # [cout][ky][kx][cin]
        arrays.append(Wk.numpy().astype(np.float32 if li == 0 else np.float16).tobytes())
        arrays.append(b.numpy().astype(np.float32).tobytes())
    for k, (cin, dim) in enumerate(zip(HEAD_INPUTS, OUTPUT_DIMS)):
        Wa = w[f"adapt{k}.weight"].detach().float().reshape(dim, cin)
        Wu = w[f"unc{k}.weight"].detach().float().reshape(1, cin)
        Wt = torch.cat([Wa, Wu], 0).t().contiguous()  # [cin][dim+1]
        bt = torch.cat([w[f"adapt{k}.bias"].float(), w[f"unc{k}.bias"].float()])
        arrays.append(Wt.numpy().astype(np.float32).tobytes())
        arrays.append(bt.numpy().astype(np.float32).tobytes())
    head = bytearray()
    head += MAGIC
    head += struct.pack("<ii", len(names), len(OUTPUT_DIMS))
    for cin, cout in dims:
        head += struct.pack("<ii", cin, cout)
    for cin, dim in zip(HEAD_INPUTS, OUTPUT_DIMS):
        head += struct.pack("<ii", cin, dim)
    table_off = len(head)
    data_off = _align16(table_off + 16 * len(arrays))
    offs, cur = [], data_off
    for a in arrays:
        offs.append((cur, len(a)))
        cur = _align16(cur + len(a))
    for o, n in offs:
        head += struct.pack("<qq", o, n)
    blob = bytearray(cur)
    blob[: len(head)] = head
    for (o, n), a in zip(offs, arrays):
        blob[o : o + n] = a
    return bytes(blob)


class UNet:
    """``model({"image": 1x3xHxW in [0,1]})`` compatible wrapper + the packed fast path."""

    scales = [1, 4, 16]
    output_dims = OUTPUT_DIMS

    def __init__(self, weights: Dict[str, torch.Tensor], device: torch.device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PxtError("UNet needs a ROCm device; no CPU path exists")
        blob = pack_unet_weights(weights)
        import hashlib

        self.weights_signature = hashlib.sha1(blob).hexdigest()  # (lock-step trackers check that they share a checkpoint)
        self._ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().pxt_unet_create(blob, len(blob), C.byref(self._ctx)), "pxt_unet_create")
        self._ws: Optional[torch.Tensor] = None

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                _lib.lib().pxt_unet_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    # pixloc module-ish surface
    def eval(self):
        return self

    def to(self, device):
        assert torch.device(device) == self.device
        return self

    @staticmethod
    def level_shapes(H: int, W: int) -> List[Tuple[int, int]]:
        hs, ws = [H], [W]
        for _ in range(4):
            hs.append(hs[-1] // 2)
            ws.append(ws[-1] // 2)
        # decoder output sizes (upsample x2 of the coarser map, skip cropped to it)
        dh, dw = hs[4], ws[4]
        dec = []
        for _ in range(4):
            dh, dw = dh * 2, dw * 2
            dec.append((dh, dw))
        return [dec[3], dec[1], (hs[4], ws[4])]  # strides 1, 4, 16

    def forward_packed(self, image: torch.Tensor, mask: Optional[torch.Tensor] = None,
                       normalize: bool = False) -> List[torch.Tensor]:
        """image: HWC, 3 channels, 0..255, float32 or uint8, on the device.  Returns the
        three HWC float32 maps [h,w,cstride] (descriptor channels then confidence)."""
        return self.forward_packed_batch([(image, mask, normalize)])[0]

    def forward_packed_batch(self, items) -> List[List[torch.Tensor]]:
        """items: [(image, mask or None, normalize)], images of one size (or exactly two of different sizes) -> per
        image the three maps of forward_packed (pxt_unet_forward_batch / pxt_unet_forward_pair)."""
        n = len(items)
        sizes = [(int(it[0].shape[0]), int(it[0].shape[1])) for it in items]
        H, W = sizes[0]
        for image, _mask, _ in items:
            _lib.require_gpu(image, "image")
        if n == 2 and sizes[1] != sizes[0]:  # two sizes: two single-image passes side by side (pxt_unet_forward_pair)
            Hs, Ws = (C.c_int32 * 2)(sizes[0][0], sizes[1][0]), (C.c_int32 * 2)(sizes[0][1], sizes[1][1])
            need = int(_lib.lib().pxt_unet_workspace_bytes_pair(self._ctx, Hs, Ws))
        else:
            need = int(_lib.lib().pxt_unet_workspace_bytes_batch(self._ctx, n, H, W))
        if need <= 0:
            raise _lib.PxtError(f"images {sizes} are not supported by the 4-level encoder")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        outs = [[torch.empty(h, w, cstride_for(c), device=self.device, dtype=torch.float32)
                 for (h, w), c in zip(self.level_shapes(*hw), OUTPUT_DIMS)] for hw in sizes]
        ops.unet_forward_batch(int(self._ctx.value), [it[0] for it in items], [it[1] for it in items],
                               [bool(it[2]) for it in items], [o for per in outs for o in per], self._ws)
        return outs

    def set_batch_plan(self, per_image_plan: bool) -> None:
        """How batches of more than two images plan their layers (pxt_unet_set_batch_plan): False = for the batch as
        launched (the fastest), True = every layer as a single image of the pair pass (each image's maps bit for bit
        those of the one-object tracker)."""
        _lib.check(_lib.lib().pxt_unet_set_batch_plan(self._ctx, int(bool(per_image_plan))), "pxt_unet_set_batch_plan")

    def set_tile_skip(self, on: bool) -> None:
        """Constant-tile skipping in encoder blocks 1-3 (pxt_unet_set_tile_skip; on by default): tiles whose dependency
        cone lies where the input is constant (masked-out query, background of the reference render) are filled with
        the layer's constant output instead of being computed - the same bits."""
        _lib.check(_lib.lib().pxt_unet_set_tile_skip(self._ctx, int(bool(on))), "pxt_unet_set_tile_skip")

    def set_defer_join(self, on: bool) -> None:
        """With True, a two-image call returns with the FIRST image's maps complete in the current stream's order and
        the second image's pass still running on the library's side stream: the caller may enqueue work on the first
        image's maps and must call join() before anything reads the second image's (pxt_unet_set_defer_join)."""
        _lib.check(_lib.lib().pxt_unet_set_defer_join(self._ctx, int(bool(on))), "pxt_unet_set_defer_join")

    def join(self) -> None:
        """Makes the current stream wait for a second pass left running by a deferred-join call (no-op otherwise)."""
        _lib.check(_lib.lib().pxt_unet_pair_join(self._ctx, _lib.stream_ptr(self.device)), "pxt_unet_pair_join")

    def activation_stats(self, image: torch.Tensor, mask: Optional[torch.Tensor] = None):
        """Range check of the fp16 activations for one image (HWC 0..255 on the device): runs a single-image pass and
        returns [(largest |activation|, number of non-finite values)] for the 17 convolutions (None for the two layers
        that are never written to memory).  pixloc runs this network in fp32; with a real checkpoint, call this on a few
        frames before trusting the poses: a count > 0 or a maximum near 65504 means fp16 storage overflows there."""
        self.forward_packed(image, mask, normalize=False)
        H, W = int(image.shape[0]), int(image.shape[1])
        stats = torch.zeros(34, dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().pxt_unet_activation_stats(self._ctx, H, W, self._ws.data_ptr(), stats.data_ptr(),
                                                        _lib.stream_ptr(self.device)), "pxt_unet_activation_stats")
        h = stats.cpu()
        counts = h.view(torch.int32)
        return [None if float(h[2 * l]) < 0 else (float(h[2 * l]), int(counts[2 * l + 1])) for l in range(17)]

    def __call__(self, data: Dict[str, torch.Tensor]) -> Dict[str, List[torch.Tensor]]:
        image = data["image"]  # 1 x 3 x H x W in [0, 1]
        assert image.dim() == 4 and image.shape[0] == 1 and image.shape[1] == 3
        hwc = (image[0].permute(1, 2, 0) * 255.0).to(self.device, torch.float32).contiguous()
        outs = self.forward_packed(hwc, None, normalize=False)
        feats = [o[..., :c].permute(2, 0, 1)[None] for o, c in zip(outs, OUTPUT_DIMS)]
        confs = [o[..., c : c + 1].permute(2, 0, 1)[None] for o, c in zip(outs, OUTPUT_DIMS)]
        return {"feature_maps": feats, "confidences": confs}
