"""N>1 host logic on CPU: world_size-2 gloo, pair sharding + match-table all-gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from imcui_hip.distributed import gather_match_tables, padded_shard_size, run_sharded, shard_bounds


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 9, 100):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - s for s, e in spans) <= padded_shard_size(n, world) if n else True


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    K = 16

    def fake_table(span):  # deterministic function of the global pair index, like a real matcher
        s, e = span
        idx = torch.arange(s, e, dtype=torch.int32)
        return torch.stack([idx * 3 + j for j in range(3 + 2 * K)], 1)

    out = run_sharded(lambda s, e: (s, e), num_pairs, fake_table)
    expect = fake_table((0, num_pairs))
    q.put((rank, bool(torch.equal(out, expect)), tuple(out.shape)))
    # also the padded path explicitly
    s, e = shard_bounds(num_pairs, rank, world)
    out2 = gather_match_tables(fake_table((s, e)), num_pairs)
    q.put((rank, bool(torch.equal(out2, expect)), tuple(out2.shape)))
    dist.destroy_process_group()


@pytest.mark.parametrize("num_pairs", [7, 8])
def test_gloo_world2_allgather(num_pairs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, num_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (num_pairs, 3 + 32) for _, _, shape in res)


def test_match_table_layout_with_and_without_stop():
    """pipeline.match_table: int32 [B, 3 + 2K] = n0, n1, stop, matches0[K], bit-cast scores[K]; SuperGlue has no
    early exit, its record carries stop = 0."""
    import torch

    from imcui_hip.pipeline import match_table

    out = {"matches0": torch.tensor([[1, -1, 0], [2, 2, -1]], dtype=torch.int32), "matching_scores0": torch.tensor([[0.5, 0.0, 0.25], [1.0, 0.125, 0.0]]),
           "num_keypoints0": torch.tensor([3, 2], dtype=torch.int32), "num_keypoints1": torch.tensor([2, 3], dtype=torch.int32)}  # fmt: skip
    t = match_table(out)
    assert t.dtype == torch.int32 and t.shape == (2, 9)
    assert t[:, :3].tolist() == [[3, 2, 0], [2, 3, 0]] and t[:, 3:6].tolist() == [[1, -1, 0], [2, 2, -1]]
    assert torch.equal(t[:, 6:].contiguous().view(torch.float32), out["matching_scores0"])
    out["stop"] = torch.tensor([9, 4], dtype=torch.int32)
    assert match_table(out)[:, 2].tolist() == [9, 4]


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher around it must run TWO ranks (it re-executes itself under
    torch.distributed.run) and report n_gpus = 2; a launcher that started a different rank count is refused.  Runs the
    launch / timing / all-gather scaffold of bench.py on gloo (`--workload launchcheck`: no HIP work)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "launchcheck", "--gpus", "2", "--steps", "3"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)  # fmt: skip
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "launchcheck", "--gpus", "4", "--steps", "1"],
                         capture_output=True, text=True, timeout=120, env={**env, "WORLD_SIZE": "2", "RANK": "0"}, cwd=root)  # fmt: skip
    assert bad.returncode != 0 and "WORLD_SIZE=2" in (bad.stderr + bad.stdout)
