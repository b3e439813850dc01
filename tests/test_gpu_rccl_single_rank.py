"""The multi-GPU launch path on hardware, as far as a one-GPU box allows: `bench.py` under the driver's own launcher line
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 ...`) with IMCUI_BENCH_DIST1=1, which makes the single rank
take every branch the N > 1 job takes -- `init_process_group("nccl", device_id=...)` (RCCL), the barriers around the timed loop, the asynchronous
`all_gather_into_tensor` of the match tables (`distributed.TableGather`), the rank-agreement reduction and the max-over-ranks reduction of the time.
World-size-2 behaviour (sharding, gather layout) is covered on CPU with gloo in tests/test_distributed_cpu.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["splg", "loftr"])
def test_bench_under_torchrun_single_rank_rccl(workload):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, IMCUI_BENCH_DIST1="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    extra = ["--batch", "4"] if workload == "splg" else ["--workload", "loftr", "--batch", "1", "--size", "256", "256"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-legs", "--no-cpu-baseline", "--no-parity", *extra]  # fmt: skip
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["scaling"] == "weak"
    assert "[bench] RCCL process group up: 1 rank(s), backend nccl" in r.stderr, r.stderr[-2000:]
