"""One shape, one configuration, N launches: the target of rocprofv3 --pmc runs.
    python scripts/bench_conv_probe.py H W Cin Cout cfg [launches]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pixtrack_amd import _lib

H, W, Cin, Cout, cfg = (int(a) for a in sys.argv[1:6])
n = int(sys.argv[6]) if len(sys.argv) > 6 else 10
dev = torch.device("cuda:0")
L = _lib.lib()
x = torch.randn(H, W, Cin, device=dev).half()
w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05).half()
b = torch.randn(Cout, device=dev)
packed = torch.empty(int(L.pxt_conv3x3_packed_bytes(Cin, Cout)), dtype=torch.uint8, device=dev)
_lib.check(L.pxt_conv3x3_pack_weights(w.data_ptr(), Cin, Cout, packed.data_ptr(), _lib.stream_ptr(dev)), "pack")
out = torch.empty(H, W, Cout, device=dev, dtype=torch.float16)
for _ in range(n):
    _lib.check(L.pxt_conv3x3_packed(x.data_ptr(), H, W, Cin, packed.data_ptr(), b.data_ptr(), Cout, 1, out.data_ptr(), None,
                                    cfg, 1, None, 0, _lib.stream_ptr(dev)), "conv")
torch.cuda.synchronize()
