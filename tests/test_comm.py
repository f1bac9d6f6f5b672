"""Multi-GPU behind the C ABI (lscqp_comm_*, lscqp_solve_batch_sharded, lscqp_allgather): one host process, G devices.
The GPU tests run with G = the number of visible devices (1 on the test box: the same code path with one block, RCCL
initialised over one device); the partition rule is tested on CPU for every G."""
import threading

import numpy as np
import pytest


def test_partition_rule_matches_the_survey_and_the_python_sharding(api):
    """Contiguous blocks of ceil(N / G) in agent order -- the C ABI's rule equals sharding.shard_range
    (the torch.distributed path of bench.py), covers [0, N) exactly once, and is ragged only in the last non-empty block."""
    from lsc_dr_planner_amd import sharding

    for N in (0, 1, 7, 8, 10, 63, 64, 512, 1024, 4096):
        for G in (1, 2, 3, 4, 8):
            blocks = [api.shard_range(N, G, g) for g in range(G)]
            assert [(f, f + c) for f, c in blocks] == [sharding.shard_range(N, G, g) for g in range(G)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == N
            assert all(blocks[g][0] + blocks[g][1] == blocks[g + 1][0] for g in range(G - 1))
            assert max(c for _, c in blocks) == -(-N // G)
    with pytest.raises(api.LscqpError):
        api.shard_range(-1, 2, 0)


@pytest.mark.gpu
def test_sharded_solve_equals_the_single_device_solve(api, oracle):
    import torch

    from lsc_dr_planner_amd import synth

    comm = api.Comm()  # every visible device
    assert comm.size == torch.cuda.device_count()
    assert "rccl" in comm.backend and "ncclCommInitAll" in comm.backend, comm.backend
    # the spreading rule: 256 agents per device by default, never more devices than the communicator has
    assert comm.devices_for(64) == 1 and comm.devices_for(255) == 1
    assert comm.devices_for(4096) == min(comm.size, 16)
    N, M, dim = 300, 5, 3
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=12, seed=9)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    # round 5: the rule lscqp_solve_batch_sharded applies knows the class -- a second device only where ONE device would need a second round
    # of workgroups in the first kernel of the solve (lscqp_device_fill: thousands of QPs with the dual active-set phase)
    fill = sol.device_fill(N, 12)
    assert fill >= 256 and comm.devices_for_class(sol, 64, 12) == 1 and comm.devices_for_class(sol, fill - 1, 12) == 1
    assert comm.devices_for_class(sol, 40 * fill, 12) == min(comm.size, 40)
    comm.set_min_agents_per_device(32)  # so that a multi-GPU box really spreads this batch (a threshold the caller names overrides the class's)
    assert comm.devices_for_class(sol, 300, 12) == comm.devices_for(300) == min(comm.size, 300 // 32)
    for step in range(2):
        b = sw.build()
        hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
        x0 = api.x_init_from_swarm(b, dim)
        G1 = sol.solve_host(hdr, rows, off, sfc, x_init=x0)
        GS = sol.solve_sharded(comm, hdr, rows, off, sfc, x_init=x0)
        assert GS["devices_used"] == min(comm.size, N // 32)
        assert (G1["status"] == 0).all() and np.array_equal(G1["status"], GS["status"])
        assert np.array_equal(G1["x"], GS["x"]) and np.array_equal(G1["obj"], GS["obj"])  # same kernels, same blocks of work: bit-identical
        assert np.array_equal(G1["info"]["iterations"], GS["info"]["iterations"])
        sw.advance(G1["x"])
    # ragged obstacle counts: the row offsets of a block are rebased
    hdr2 = hdr.copy()
    hdr2["n_obs"][::3] = 0
    G1 = sol.solve_host(hdr2, rows, off, sfc)
    GS = sol.solve_sharded(comm, hdr2, rows, off, sfc)
    assert np.array_equal(G1["x"], GS["x"]) and np.array_equal(G1["status"], GS["status"])
    comm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(row_format=1), dict(precision=1), dict(row_format=1, precision=1)], ids=["rows_f32", "mixed", "mixed+rows_f32"])
def test_sharded_solve_follows_the_handles_row_format_and_precision(api, kw):
    """The sharded entry stages rows in the handle's storage format (16-byte rows are cut at 16-byte offsets) and runs the handle's
    precision mode, second pass included: identical to the single-device call."""
    from lsc_dr_planner_amd import synth

    comm = api.Comm()
    comm.set_min_agents_per_device(32)
    N, M, dim = 130, 5, 3
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=9, seed=31)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, **kw))
    b = sw.build()
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
    hdr["n_obs"][::4] = 5  # ragged: the offsets stay, fewer obstacles are read
    x0 = api.x_init_from_swarm(b, dim)
    G1 = sol.solve_host(hdr, rows, off, sfc, x_init=x0)
    GS = sol.solve_sharded(comm, hdr, rows, off, sfc, x_init=x0)
    assert (G1["status"] == 0).all() and np.array_equal(G1["status"], GS["status"])
    assert np.array_equal(G1["x"], GS["x"]) and np.array_equal(G1["obj"], GS["obj"]) and np.array_equal(G1["info"]["flags"], GS["info"]["flags"])
    comm.close()


@pytest.mark.gpu
def test_allgather_of_solved_trajectories_over_rccl(api):
    """lscqp_allgather: every device ends up with every block, in device order (broadcastMsgs' device analogue)."""
    import torch

    comm = api.Comm()
    G, count = comm.size, 64 * 90
    send, recv = [], []
    for g in range(G):
        dev = torch.device("cuda", g)
        send.append((torch.arange(count, dtype=torch.float64, device=dev) + 1000.0 * g).contiguous())
        recv.append(torch.full((G * count,), -1.0, dtype=torch.float64, device=dev))
    torch.cuda.synchronize()
    comm.allgather(send, recv, count)
    comm.synchronize()
    want = torch.cat([torch.arange(count, dtype=torch.float64) + 1000.0 * g for g in range(G)])
    for g in range(G):
        assert torch.equal(recv[g].cpu(), want)
    comm.close()


@pytest.mark.gpu
def test_host_entry_points_are_thread_safe_per_handle(api, oracle):
    """Two threads planning different agents through the SAME solver handle (the map / solver handles are shared by all agents'
    objects in the planner): each call stages through its own slot and stream, results equal the sequential ones."""
    from lsc_dr_planner_amd import synth

    N, M, dim = 64, 5, 3
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=10, seed=21)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    b = sw.build()
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
    rows2 = rows.reshape(N, -1)
    ref = sol.solve_host(hdr, rows, off, sfc)
    out = {}

    def work(tid):
        res = []
        for rep in range(20):
            for q in range(tid, N, 4):
                r = sol.solve_host(hdr[q:q + 1], rows2[q], np.array([0, rows2.shape[1]], dtype=np.uint64), sfc[q:q + 1], want_info=False)
                res.append((q, r["x"][0].copy(), int(r["status"][0])))
        out[tid] = res

    ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for tid in range(4):
        assert len(out[tid]) == 20 * (N // 4)
        for q, x, st in out[tid]:
            # (a single-QP launch takes the two-wavefront instance, the 64-QP reference too: bit-identical)
            assert st == ref["status"][q] and np.array_equal(x, ref["x"][q]), (tid, q)
