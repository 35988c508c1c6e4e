"""CPU, world_size 2, gloo: the N>1 path (file sharding + final gather of score rows)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from apr_amd import sweep


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_scores(file_id, n, C=4):
    # deterministic stand-in for Engine.score (the sharding/gather logic is what is under test)
    r = np.random.default_rng(file_id)
    p = r.random((n, C))
    p /= p.sum(axis=1, keepdims=True)
    return p, p.argmax(axis=1).astype(np.int32)


def _worker(rank, world, port, n_files, seconds, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = [s * 96000 for s in seconds]
    counts = []
    for r in range(world):
        files = sweep.shard_files(n_files, world, r)
        counts.append(sweep.plan_stream([lens[f] for f in files], files).n_windows)
    mine = sweep.shard_files(n_files, world, rank)
    plan = sweep.plan_stream([lens[f] for f in mine], mine)
    probs = np.zeros((plan.n_windows, 4))
    label = np.zeros(plan.n_windows, np.int32)
    for f in mine:
        idx = np.nonzero(plan.file_ids == f)[0]
        p, l = _fake_scores(f, len(idx))
        probs[idx], label[idx] = p, l
    pl, ll = sweep.gather_scores(torch.from_numpy(probs), torch.from_numpy(label), counts, dist)
    if rank == 0:
        q.put(([p.numpy() for p in pl], [l.numpy() for l in ll], counts))
    else:
        assert pl is None and ll is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    world, n_files = 2, 5
    seconds = [4, 3, 6, 2, 5]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_files, seconds, q)) for r in range(world)]
    for p in procs:
        p.start()
    pl, ll, counts = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # rank r's rows come back in rank r's deterministic stream order
    for r in range(world):
        files = sweep.shard_files(n_files, world, r)
        plan = sweep.plan_stream([seconds[f] * 96000 for f in files], files)
        assert pl[r].shape == (plan.n_windows, 4) and counts[r] == plan.n_windows
        for f in files:
            idx = np.nonzero(plan.file_ids == f)[0]
            p, l = _fake_scores(f, len(idx))
            assert np.array_equal(pl[r][idx], p) and np.array_equal(ll[r][idx], l)
    # every valid window of every file is present exactly once
    total_valid = sum(s - 1 for s in seconds)
    got_valid = sum(sweep.plan_stream([seconds[f] * 96000 for f in sweep.shard_files(n_files, world, r)],
                                      sweep.shard_files(n_files, world, r)).n_valid for r in range(world))
    assert got_valid == total_valid
