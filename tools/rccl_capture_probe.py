"""GPU-box aid: can this stack capture an RCCL all-reduce into a HIP graph (torch.cuda.graph)?  One rank is enough to answer (the collective
goes through ncclAllReduce either way).  usage: python tools/rccl_capture_probe.py   (MASTER_ADDR/PORT default to 127.0.0.1:29571)"""
import os
import sys
import traceback

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.ones(1 << 20, device="cuda")
dist.all_reduce(x)                      # eager once: communicator set-up is not capturable
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            y = x * 2.0
            dist.all_reduce(y, op=dist.ReduceOp.AVG)
            z = y + 1.0
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("captured + replayed: z[0] = %.1f (expect 3.0)" % float(z[0]))
except Exception:
    print("capture FAILED:")
    traceback.print_exc()
dist.destroy_process_group()
