"""world_size-2 gloo test (CPU) of the multi-GPU path: contiguous agent blocks + all-gather of solved trajectories.
The per-rank solver is stubbed by the CPU oracle here; on GPUs it is lscqp_solve_batch_device (bench.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, N, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lsc_dr_planner_amd import sharding, synth
    from oracle import oracle as O
    from tests import helpers as H

    M, dim = 5, 3
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=4, seed=5)      # every rank builds the same swarm, solves its block
    cls = O.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    b = sw.build()
    ag, lsc, off, sfc = H.swarm_oracle_inputs(O, sw, b)
    lo, hi = sharding.shard_range(N, world, rank)
    R = O.solve_batch(cls, ag[lo:hi], lsc, off[lo:hi], sfc[lo * M:], threads=1)
    x_all = sharding.allgather_trajectories(torch.from_numpy(R["x"]), N)
    np.save(os.path.join(out_dir, "x_%d.npy" % rank), x_all.numpy())
    # safety figures of the step: per-agent records of the local block (oracle here, lscqp_safety_metrics_device on GPUs),
    # then one MIN / MAX all-reduce
    agl = ag[lo:hi].copy()
    agl["vmax"][:, 0] = 0.002
    S = O.safety_metrics(cls, agl, x_all.numpy(), sw.radius, sw.downwash, 2, 0.05, first=lo)
    red = sharding.reduce_safety_metrics(torch.from_numpy(S[:, 0].copy()), torch.from_numpy(S[:, 3:6].copy()), torch.from_numpy(S[:, 6:9].copy()))
    np.save(os.path.join(out_dir, "s_%d.npy" % rank), np.concatenate([red[0].reshape(1).numpy(), red[1].numpy(), red[2].numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges():
    from lsc_dr_planner_amd import sharding

    for N in (1, 7, 8, 10, 64, 4096):
        for G in (1, 2, 4, 8):
            blocks = [sharding.shard_range(N, G, r) for r in range(G)]
            assert blocks[0][0] == 0 and blocks[-1][1] == N
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(G - 1))
            assert max(h - l for l, h in blocks) == -(-N // G)


def test_two_rank_gloo_allgather(tmp_path, oracle):
    N = 7  # ragged: blocks of 4 and 3
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, N, str(tmp_path)), nprocs=2, join=True)
    x0, x1 = np.load(tmp_path / "x_0.npy"), np.load(tmp_path / "x_1.npy")
    assert x0.shape == (N, 90) and np.array_equal(x0, x1)
    # equals the single-process solve of the whole batch
    sys.path.insert(0, ROOT)
    from lsc_dr_planner_amd import synth
    from tests import helpers as H

    sw = synth.Swarm(N, M=5, dim=3, n_obs=4, seed=5)
    cls = oracle.make_class(M=5, dim=3, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    ag, lsc, off, sfc = H.swarm_oracle_inputs(oracle, sw, sw.build())
    R = oracle.solve_batch(cls, ag, lsc, off, sfc, threads=2, tol=1e-11, max_iter=200, polish=False)  # (the worker ranks call the module with its defaults)
    assert np.array_equal(R["x"], x0)
    # the reduced safety figures equal the single-process figures over all agents, on both ranks
    s0, s1 = np.load(tmp_path / "s_0.npy"), np.load(tmp_path / "s_1.npy")
    ag["vmax"][:, 0] = 0.002
    S = oracle.safety_metrics(cls, ag, R["x"], sw.radius, sw.downwash, 2, 0.05)
    want = np.concatenate([[S[:, 0].min()], S[:, 3:6].max(axis=0), S[:, 6:9].max(axis=0)])
    assert np.array_equal(s0, s1) and np.array_equal(s0, want) and want[1] > 0


def _exchange_worker(rank, world, port, n_total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lsc_dr_planner_amd import sharding

    lo, hi = sharding.shard_range(n_total, world, rank)
    bufs = []
    for per in (120, 9, 3):  # plans (dim * M * 6), states, goal points
        b = torch.full((n_total * per,), -1.0, dtype=torch.float64)   # stale copies of the others' blocks
        own = torch.arange(lo * per, hi * per, dtype=torch.float64) + 1000.0 * per
        b[lo * per:hi * per] = own                                    # the block this rank has just replanned
        bufs.append(b)
    sharding.exchange_plan_buffers(bufs, n_total)
    np.save(os.path.join(out_dir, "b_%d.npy" % rank), torch.cat(bufs).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_exchange_of_plan_buffers(tmp_path):
    """The chain's exchange step between ranks (sharding.exchange_plan_buffers; behind the C ABI: lscqp_plan_group_step): after it every
    rank holds every owner's block of the plan / state / goal buffers, ragged blocks included (7 agents: 4 + 3)."""
    n_total = 7
    port = 31500 + os.getpid() % 2000
    mp.spawn(_exchange_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    b0, b1 = np.load(tmp_path / "b_0.npy"), np.load(tmp_path / "b_1.npy")
    assert np.array_equal(b0, b1)
    want = np.concatenate([np.arange(n_total * per, dtype=np.float64) + 1000.0 * per for per in (120, 9, 3)])
    assert np.array_equal(b0, want)


def _apply_exchange(bufs, ops, first, per):
    """What RCCL does with the operations of lscqp_exchange_schedule, on host arrays (one per device)."""
    from lsc_dr_planner_amd import api

    G = len(bufs)
    for op in ops:
        if op["kind"] == api.XCHG_ALLGATHER:  # in place: device g sends `count` doubles from g * count (its own block), block o lands at o * count
            sent = [bufs[g][g * op["count"]: (g + 1) * op["count"]].copy() for g in range(G)]
            assert all(len(s) == op["count"] for s in sent), "an all-gather block reaches beyond a device's buffer"
            for g in range(G):
                for o in range(G):
                    bufs[g][o * op["count"]: (o + 1) * op["count"]] = sent[o]
        else:
            src = bufs[op["root"]][op["offset"]: op["offset"] + op["count"]].copy()
            for g in range(G):
                bufs[g][op["offset"]: op["offset"] + op["count"]] = src


def test_exchange_schedule_delivers_every_owners_block_for_any_device_count(api):
    """The bookkeeping of the sharded replan's exchange (lscqp_plan_group_step / the bench's all-gather) for G in {2, 3, 4, 8} devices and
    agent counts that split evenly, raggedly, and leave devices EMPTY -- without a device: lscqp_shard_range cuts the mission,
    lscqp_exchange_schedule turns the blocks into collectives, the collectives are played on host arrays: afterwards every device
    holds every owner's block, nothing out of bounds, an empty owner never roots a broadcast."""
    rng = np.random.default_rng(3)
    for G in (1, 2, 3, 4, 8):
        for n in sorted({1, 2, G - 1, G, G + 1, 9, 10, 64, 512, 1000, 1024, 4096} - {0}):
            for per in (1, 3, 90):
                blocks = [api.shard_range(n, G, g) for g in range(G)]
                first, count = [b[0] for b in blocks], [b[1] for b in blocks]
                assert first[0] == 0 and sum(count) == n and all(first[g + 1] == first[g] + count[g] for g in range(G - 1))
                assert max(count) == -(-n // G)  # contiguous blocks of ceil(N / G), SURVEY.md section 8e
                ops = api.exchange_schedule(n, first, count, per)
                truth = rng.standard_normal(n * per)
                bufs = []
                for g in range(G):  # device g holds stale data everywhere but in its own, freshly written block
                    b = rng.standard_normal(n * per)
                    b[first[g] * per: (first[g] + count[g]) * per] = truth[first[g] * per: (first[g] + count[g]) * per]
                    bufs.append(b)
                equal = len(set(count)) == 1
                if equal:
                    assert len(ops) == 1 and ops[0]["kind"] == api.XCHG_ALLGATHER and ops[0]["count"] == count[0] * per
                else:
                    assert all(o["kind"] == api.XCHG_BROADCAST for o in ops) and [int(o["root"]) for o in ops] == [g for g in range(G) if count[g] > 0]
                    for o in ops:
                        assert o["offset"] == first[o["root"]] * per and o["count"] == count[o["root"]] * per and o["offset"] + o["count"] <= n * per
                _apply_exchange(bufs, ops, first, per)
                for g in range(G):
                    assert np.array_equal(bufs[g], truth), (G, n, per, g)
    # blocks that do not tile the mission are refused
    import pytest

    for first, count in (([0, 5], [4, 5]), ([0, 4], [4, 5]), ([1, 5], [4, 5])):
        with pytest.raises(api.LscqpError):
            api.exchange_schedule(10, first, count, 1)


def test_a_ragged_mission_is_one_allgather_when_the_buffers_carry_padding(api):
    """lscqp_exchange_schedule_padded: with LSCQP_PLAN_EXCHANGE_PAD agents of room behind the mission (what lscqp_plan_create allocates for the
    buffers a group exchanges) the blocks lscqp_shard_range cuts are ONE in-place all-gather for EVERY agent count -- short and empty
    blocks send from and into the padding; played on host arrays the mission part of every device's buffer ends up complete, and nothing
    is touched beyond the padding.  Without enough padding, or for blocks of another shape, the broadcasts."""
    rng = np.random.default_rng(5)
    pad = api.PLAN_EXCHANGE_PAD
    for G in (1, 2, 3, 4, 8):
        for n in sorted({1, 2, G - 1, G, G + 1, 9, 10, 63, 64, 65, 512, 1000, 1023, 4096} - {0}):
            for per in (1, 9, 90):
                blocks = [api.shard_range(n, G, g) for g in range(G)]
                first, count = [b[0] for b in blocks], [b[1] for b in blocks]
                ops = api.exchange_schedule(n, first, count, per, pad_agents=pad)
                B = -(-n // G)
                assert len(ops) == 1 and ops[0]["kind"] == api.XCHG_ALLGATHER and ops[0]["count"] == B * per, (G, n, ops)
                assert G * B - n <= pad
                truth = rng.standard_normal(n * per)
                bufs = []
                for g in range(G):
                    b = rng.standard_normal((n + pad) * per)
                    b[first[g] * per: (first[g] + count[g]) * per] = truth[first[g] * per: (first[g] + count[g]) * per]
                    bufs.append(b)
                _apply_exchange(bufs, ops, first, per)
                for g in range(G):
                    assert np.array_equal(bufs[g][: n * per], truth), (G, n, per, g)
    # not enough room behind the mission: the broadcasts of the unpadded entry
    first, count = zip(*[api.shard_range(9, 8, g) for g in range(8)])  # 2,2,2,2,1,0,0,0: 8 * 2 - 9 = 7 agents of padding needed
    assert len(api.exchange_schedule(9, first, count, 3, pad_agents=7)) == 1
    ops = api.exchange_schedule(9, first, count, 3, pad_agents=6)
    assert len(ops) == 5 and all(o["kind"] == api.XCHG_BROADCAST for o in ops)
    # blocks of another shape (not what lscqp_shard_range cuts): broadcasts whatever the padding
    ops = api.exchange_schedule(10, [0, 3, 7], [3, 4, 3], 2, pad_agents=pad)
    assert len(ops) == 3 and all(o["kind"] == api.XCHG_BROADCAST for o in ops)

