import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import numpy as np
from monocular_slam_amd import api, synth
from monocular_slam_amd.tracking import TrackingBatch
ctx = api.Context(0, 640, 480, 2048, 2)
tb = TrackingBatch(ctx, 2, 640, 480, 500)
tb.upload_frames(np.stack([synth.textured_frame(640, 480, 1), synth.textured_frame(640, 480, 2)]))
tb.step(); tb.step(); tb.sync()
dist.barrier(); torch.cuda.synchronize()
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("nccl + mslam in one process ok:", float(t.item()), tb.results()["n"])
dist.destroy_process_group()
