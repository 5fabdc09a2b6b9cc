"""torch.library registration of the C ABI: every hot-path entry point of
include/pixtrack_hip.h is a `torch.ops.pixtrack.*` op for the ROCm device only (no CPU kernel)."""
import pytest
import torch

from pixtrack_amd import _lib, ops


def test_every_hot_path_symbol_has_an_op():
    names = ops.op_names()
    for sym in ("pxt_lm_refine", "pxt_sample_sparse", "pxt_unet_forward_batch", "pxt_ngp_render",
                "pxt_ngp_render_both", "pxt_depth_mask", "pxt_rgba_to_u8", "pxt_resize_linear",
                "pxt_conv3x3_nhwc_f16"):
        assert sym in _lib.PROTOTYPES
        assert sym[len("pxt_"):] in names
        assert hasattr(torch.ops.pixtrack, sym[len("pxt_"):])


def test_schemas_are_out_variants_and_cuda_only():
    s = str(torch.ops.pixtrack.lm_refine.default._schema)
    assert "Tensor(a!) record" in s and "Tensor[] fmaps" in s
    s = str(torch.ops.pixtrack.ngp_render_both.default._schema)
    assert "Tensor(a!) rgba" in s and "Tensor(b!) depth" in s
    # no CPU kernel anywhere: the dispatcher refuses host tensors instead of falling back
    with pytest.raises(NotImplementedError):
        torch.ops.pixtrack.rgba_to_u8(torch.zeros(4, 4, 4), 0.0, torch.zeros(4, 4, 3, dtype=torch.uint8))
    with pytest.raises(NotImplementedError):
        torch.ops.pixtrack.depth_mask(torch.zeros(4, 4, 4), 1, 5, torch.zeros(4, 4, dtype=torch.uint8),
                                      torch.zeros(32, dtype=torch.uint8))
