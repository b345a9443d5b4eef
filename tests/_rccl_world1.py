"""Helper of tests/test_gpu_rccl.py (not a test module): one rank, backend "nccl" (= RCCL on ROCm).  Initialises the process group exactly as bench.py does,
runs the probe all_reduce and the ONE collective of the sharded path (an all_gather_into_tensor of the int32 [rows, 1 + width] token block,
star-vector_amd/parallel.py) on device tensors, and prints a JSON line.  A world of one has no peer, so nothing crosses xGMI -- what this executes is RCCL
itself on this box: library load, communicator creation, the HSA IPC environment, stream-ordered device collectives."""
import datetime
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=120))
world = dist.get_world_size()
probe = torch.ones(1, device=dev)
dist.all_reduce(probe)
torch.cuda.synchronize()
block = torch.arange(32 * (1 + 1026), dtype=torch.int32, device=dev).view(32, 1027)          # BASELINE config 2's block: 32 rows, 2 prompt ids + 1024 new tokens
out = torch.empty((world * 32, 1027), dtype=torch.int32, device=dev)
dist.all_gather_into_tensor(out, block)
dist.barrier()
torch.cuda.synchronize()
try:
    ver = ".".join(str(x) for x in torch.cuda.nccl.version())
except Exception:
    ver = "unknown"
print(json.dumps({"world": world, "all_reduce": float(probe.item()), "all_gather_equal": bool(torch.equal(out[:32], block)), "rccl_version": ver,
                  "backend": dist.get_backend()}))
dist.destroy_process_group()
