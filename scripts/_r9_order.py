import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pixtrack_amd.synthetic import REF_CAMERA_PHONE
dev = torch.device("cuda:0")
def r9():
    r = bench.r9_refshape_extra(dev, REF_CAMERA_PHONE, "phone")
    return r["frames_per_s"]
print("r9 first:", r9(), flush=True)
print("r9 again:", r9(), flush=True)
print("ycb_policy:", bench.ycb_policy_extra(dev)["frames_per_s"], flush=True)
print("r9 after ycb_policy:", r9(), flush=True)
print("ycb_refshape:", bench.ycb_policy_extra(dev, refshape=True)["frames_per_s"], flush=True)
print("r9 after ycb_refshape:", r9(), flush=True)
gc.collect(); torch.cuda.empty_cache()
print("r9 after gc + empty_cache:", r9(), flush=True)
