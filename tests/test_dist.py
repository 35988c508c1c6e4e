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
    match = ((label != 2) & (probs.max(axis=1) > 0.3)).astype(np.uint8)
    pl, ll, ml = sweep.gather_scores(torch.from_numpy(probs), torch.from_numpy(label), torch.from_numpy(match), counts, dist)
    if rank == 0:
        q.put(([p.numpy() for p in pl], [l.numpy() for l in ll], [m.numpy() for m in ml], counts))
    else:
        assert pl is None and ll is None and ml is None
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
    pl, ll, ml, counts = q.get(timeout=120)
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
            assert np.array_equal(ml[r][idx], ((l != 2) & (p.max(axis=1) > 0.3)).astype(np.uint8))
    # every valid window of every file is present exactly once
    total_valid = sum(s - 1 for s in seconds)
    got_valid = sum(sweep.plan_stream([seconds[f] * 96000 for f in sweep.shard_files(n_files, world, r)],
                                      sweep.shard_files(n_files, world, r)).n_valid for r in range(world))
    assert got_valid == total_valid


def test_row_packing_is_byte_exact():
    rng = np.random.default_rng(1)
    p = torch.from_numpy(rng.random((7, 4)))
    l = torch.from_numpy(np.array([0, 3, -1, 2, 1, 0, 3], np.int32))
    m = torch.from_numpy(np.array([1, 0, 0, 1, 1, 0, 1], np.uint8))
    rows = sweep.pack_rows(p, l, m, cap=9)
    assert rows.shape == (9, sweep.row_bytes(4)) and sweep.row_bytes(4) == 37 and not rows[7:].any()
    p2, l2, m2 = sweep.unpack_rows(rows[:7], 4)
    assert torch.equal(p, p2) and torch.equal(l, l2) and torch.equal(m, m2)


def test_plan_shards_covers_every_recording_once():
    for world in (1, 2, 4, 8):
        plans = [sweep.plan_shards(61, 5, world, r) for r in range(world)]
        assert sorted(sum((p.files for p in plans), [])) == list(range(61))
        assert all(p.counts == plans[0].counts and p.valid_total == 61 * 4 for p in plans)
        assert [p.plan.n_windows for p in plans] == plans[0].counts
    # BASELINE config 3: 6000 recordings x 600 s over 8 GPUs
    p = sweep.plan_shards(6000, 600, 8, 3)
    assert len(p.files) == 750 and p.files[:2] == [3, 11] and p.valid_total == 3594000 and p.plan.n_valid == 750 * 599


def _bench(*argv, env=None):
    import json
    import subprocess
    import sys
    from conftest import REPO
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_launches_its_own_ranks_dry_run():
    # `python bench.py --gpus 2` (no torchrun): two fresh rank processes, gloo, stand-in scorer; plan + gather verified on rank 0
    rc, out, err = _bench("--gpus", "2", "--dry-run", "--config", "3", "--hours", "1", "--steps", "2", "--warmup", "1")
    assert rc == 0, err
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["value"] is None and out["scaling"] == "strong"
    assert out["config"]["recordings_total"] == 6 and out["config"]["windows_total"] == 6 * 599
    assert out["gather"] == {"rows": [1799, 1799], "row_bytes": 37, "bytes_per_step": 2 * 1799 * 37, "verified": True}
    # weak-scaling mode: config 2 per GPU
    rc, out, err = _bench("--gpus", "2", "--dry-run", "--files", "3", "--seconds", "7", "--steps", "1", "--warmup", "0")
    assert rc == 0, err
    assert out["scaling"] == "weak" and out["config"]["recordings_total"] == 6 and out["config"]["windows_total"] == 36


def test_bench_refuses_a_world_that_is_not_gpus():
    rc, out, err = _bench("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "3", "RANK": "0"})
    assert rc != 0 and out is None and "WORLD_SIZE=3 but --gpus 2" in err


def test_plan_clips_contiguous_blocks_by_index():
    # "Cfg 5: by clip index"; 8(d): 10 000 clips, 1 250 per GPU at 8
    for world in (1, 2, 4, 8):
        plans = [sweep.plan_clips(10000, world, r) for r in range(world)]
        assert [p.n_local for p in plans] == [10000 // world] * world == plans[0].counts
        assert plans[0].lo == 0 and plans[-1].hi == 10000 and all(a.hi == b.lo for a, b in zip(plans, plans[1:]))
    plans = [sweep.plan_clips(10, 4, r) for r in range(4)]                      # ragged: every clip exactly once, sizes differ by <= 1
    assert [(p.lo, p.hi) for p in plans] == [(0, 2), (2, 5), (5, 7), (7, 10)] and plans[2].counts == [2, 3, 2, 3]
    assert sweep.plan_clips(0, 3, 1).n_local == 0 and sweep.plan_clips(2, 8, 7).counts == [0, 0, 0, 1, 0, 0, 0, 1]
    import pytest
    with pytest.raises(ValueError):
        sweep.plan_clips(10, 2, 2)


def test_bench_config5_dry_run_shards_by_clip_index():
    for world in (1, 2, 4, 8):
        rc, out, err = _bench("--gpus", str(world), "--dry-run", "--config", "5", "--steps", "1", "--warmup", "0")
        assert rc == 0, err
        assert out["n_gpus"] == world and out["dry_run"] is True and out["value"] is None and out["unit"] == "clips/s"
        assert out["config"]["baseline_config"] == 5 and out["config"]["clips_per_rank"] == [10000 // world] * world
    # with the optional gather: rank 0 ends up with every clip's row, in data-set order
    rc, out, err = _bench("--gpus", "2", "--dry-run", "--config", "5", "--clips", "11", "--gather", "--steps", "2", "--warmup", "1")
    assert rc == 0, err
    assert out["config"]["clips_per_rank"] == [5, 6] and out["gather"] == {"rows": [5, 6], "bytes_per_step": 11 * 6 * 4, "verified": True}
