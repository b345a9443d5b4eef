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
    # one rank per GPU does the same work on the same clock: a straggler (a throttled GPU, a rank that fell back to a slower path)
    # must be REPORTED by the first real multi-GPU run, not averaged away
    by_rank = res["config"]["tokens_per_s_by_rank"]
    assert len(by_rank) == 2 and min(by_rank) > 0
    assert (max(by_rank) - min(by_rank)) / max(by_rank) < 0.05, f"per-rank throughput spread above 5 %: {by_rank}"


def test_rccl_itself_runs_on_this_box_world_of_one():
    """What a 1-GPU box CAN execute of the multi-GPU path: backend "nccl" (RCCL) initialised as bench.py initialises it, the probe all_reduce and the
    sharded path's one collective (all_gather_into_tensor of the token block) on device tensors -- with a world of one (tests/_rccl_world1.py).  No peer,
    so nothing crosses xGMI; but the library, the communicator, the IPC environment and stream-ordered device collectives have then run on real hardware
    before the driver's 8-GPU bench does."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_rccl_world1.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, f"RCCL world-of-one run failed (rc {r.returncode}):\n{r.stderr[-3000:]}"
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print(f"[rccl world 1] {res}")
    assert res["world"] == 1 and res["backend"] == "nccl" and res["all_reduce"] == 1.0 and res["all_gather_equal"]


def test_bench_bare_invocation_launches_itself_two_ranks_share_the_gpu():
    """`python bench.py --gpus 2` with NO launcher in the environment (what a driver without torchrun would type) must become the
    torch.distributed.run job itself.  On this 1-GPU box the two ranks share the device and rendezvous over gloo: the sharded path, the
    single all_gather and the line's multi-rank fields run for real; with >= 2 GPUs the same command goes over RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SV_DIST_TIMEOUT_S"] = "120"
    if torch.cuda.device_count() < 2:
        env["SV_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--new-tokens", "8",
           "--ttft-requests", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"bare bench.py --gpus 2 failed (rc {r.returncode}):\n{r.stderr[-3000:]}"
    assert "re-executing as" in r.stderr
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 64 and res["config"]["collectives_per_step"] == 1
    assert len(res["config"]["tokens_per_s_by_rank"]) == 2 and min(res["config"]["tokens_per_s_by_rank"]) > 0
    assert res["value"] > 0
