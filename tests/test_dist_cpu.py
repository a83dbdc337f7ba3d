"""CPU, world_size 2 over gloo: utterance sharding, arena broadcast and hypothesis gather."""
import os

import pytest

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG_NAME, ROOT, sub


def test_sharding_helpers():
    d = sub("dist")
    assert [d.shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [d.shard_bounds(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    parts = d.shard_by_length([30, 8, 8, 8, 5, 1], 2)
    assert sorted(sum(parts, [])) == list(range(6))
    loads = [sum([30, 8, 8, 8, 5, 1][i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 1
    tok = np.array([[5, 6, 7, 0], [9, 0, 0, 0]], np.int32)
    slab = d.pack_hypotheses(tok, np.array([3, 1], np.int32), 6)
    assert slab.shape == (2, 7)
    hyp = d.unpack_hypotheses(slab)
    assert hyp[0].tolist() == [5, 6, 7] and hyp[1].tolist() == [9]


def _worker(rank, world, port, out_dir):
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    d = importlib.import_module(PKG_NAME + ".dist")
    r, lr, w = d.init_from_env("gloo")
    assert (r, w) == (rank, world)
    dev = torch.device("cpu")
    blob = (np.arange(100003, dtype=np.int64) % 251).astype(np.uint8) if rank == 0 else None
    t = d.broadcast_arena(blob, dev)
    assert t.dtype == torch.uint8 and t.numel() == 100003 and int(t[100002]) == 100002 % 251
    # each rank "transcribes" its shard of 5 utterances: utterance i -> tokens [i, i+1, ..., ] of length i+1
    lo, hi = d.shard_bounds(5, rank, world)
    width = 8
    tok = np.zeros((3, width), np.int32)      # fixed slab height (max shard size) so all ranks send equal shapes
    num = np.zeros((3,), np.int32)
    for j, i in enumerate(range(lo, hi)):
        num[j] = i + 1
        tok[j, :i + 1] = np.arange(i, 2 * i + 1)
    got = d.gather_hypotheses(d.pack_hypotheses(tok, num, width), dev)
    if rank == 0:
        hyps = []
        for rk, slab in enumerate(got):
            l, h = d.shard_bounds(5, rk, world)
            hyps += d.unpack_hypotheses(slab)[: h - l]
        assert [x.tolist() for x in hyps] == [list(range(i, 2 * i + 1)) for i in range(5)]
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    else:
        assert got is None
    # bench.py's world > 1 legs: the per-step hypothesis slab of a full batch (B_local x (1 + max_T)) and the max-over-ranks clock
    B_local, max_t = 64, 137
    tok2 = np.full((B_local, max_t), rank + 1, np.int32)
    num2 = np.full((B_local,), 5 + rank, np.int32)
    got2 = d.gather_hypotheses(d.pack_hypotheses(tok2, num2, max_t), dev)
    if rank == 0:
        assert len(got2) == world and all(s.shape == (B_local, 1 + max_t) for s in got2)
        assert [int(s[0, 0]) for s in got2] == [5, 6] and [int(s[3, 1]) for s in got2] == [1, 2]
    assert d.max_over_ranks(1.0 + rank, dev) == float(world)
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_gather_world2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").read_text() == "ok"


@pytest.mark.parametrize("workload", ["sensevoice", "whisper", "mixed"])
def test_bench_self_launches_under_torchrun(workload):
    """`python bench.py --workload W --gpus 2` with no launcher in the environment re-executes itself under torch.distributed.run (127.0.0.1 rendezvous):
    two ranks start, meet at the barrier, rank 0 prints ONE JSON line with n_gpus = 2 (ASR_BENCH_DRYRUN=1: the step loop's launch / clock path
    on gloo, no GPU work: arena broadcast from rank 0, one hypothesis gather per step, slowest rank's clock). whisper = BASELINE.json configs[3]'s
    launch line, mixed = configs[4]'s. The round-2 bench died on `assert world == args.gpus` here."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ASR_BENCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", workload, "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["workload"] == workload
    B = 32 if workload == "whisper" else 64
    assert rec["global_batch"] == 2 * B and rec["gathered_hypotheses"] == 3 * 2 * B and rec["arena_broadcast_bytes"] == 4 << 16


def test_foreign_section_is_a_no_op_on_cpu_devices():
    """dist.foreign_section brackets RCCL collectives on CUDA devices (the native library's cluster kernels must not run beside them); the gloo
    path of the CPU tests passes straight through without touching the native library."""
    import torch
    d = sub("dist")
    ran = []
    with d.foreign_section(torch.device("cpu")):
        ran.append(1)
    with d.foreign_section("cpu"):
        ran.append(2)
    assert ran == [1, 2]


def test_bench_self_launch_command_shape():
    import importlib.util, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    cmd = b.self_launch_command(4, ["--gpus", "4", "--steps", "7"], port=12345)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "12345" and cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
