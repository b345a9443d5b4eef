"""GPU: the first RCCL run must not be the driver's scaling bench.  With >= 2 GPUs visible this launches `bench.py --gpus 2`
under torch.distributed.run exactly as the driver does (one rank per GPU, backend "nccl" = RCCL, rendezvous on 127.0.0.1) on
a short workload and checks the line; on a 1-GPU box it is skipped (there the same sharded path is covered functionally by the
gloo run of tests/test_parallel_gloo.py and by `bench.py --gpus 2` with both ranks sharing the GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL over xGMI)")
def test_bench_two_ranks_over_rccl():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SV_DIST_TIMEOUT_S="120")
    env.pop("SV_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--new-tokens", "48", "--ttft-requests", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"bench.py --gpus 2 failed (rc {r.returncode}):\n{r.stderr[-3000:]}"
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 64 and res["scaling"] == "weak"
    assert res["config"]["dist_backend"].startswith("nccl (RCCL") and res["config"]["collectives_per_step"] == 1
    assert res["value"] > 0 and res["config"]["hipgraph_decode"]
