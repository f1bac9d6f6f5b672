"""Multi-GPU sharding of one replan step: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm).

The N QPs of a step are independent (reference src/multi_sync_simulator.cpp:354-362), so agents are split into
contiguous blocks of ceil(N/G) in mission order (src/mission.cpp:140-153) with NO collective on the solve path.
The only exchange the reference has is broadcastMsgs (src/multi_sync_simulator.cpp:305-352: every agent receives its
neighbours' previous trajectories before the next step); its device analogue is one all-gather of the solved control
points, dim*M*6 doubles per agent, latency-bound on xGMI.  Works on CPU tensors with gloo too,
which is how it is tested without GPUs.
"""
import torch
import torch.distributed as dist


def shard_range(n_agents, world, rank):
    """[lo, hi) of the contiguous block owned by `rank`."""
    per = -(-n_agents // world)
    lo = min(rank * per, n_agents)
    return lo, min(lo + per, n_agents)


def allgather_trajectories(x_local, n_agents, group=None):
    """x_local: (n_local, nv) tensor of this rank's solved control points -> (n_agents, nv) on every rank.
    Shards may be ragged (the last block is shorter): blocks are padded to ceil(N/G) for the collective."""
    world = dist.get_world_size(group)
    per = -(-n_agents // world)
    nv = x_local.shape[1]
    send = x_local
    if x_local.shape[0] != per:
        send = torch.zeros((per, nv), dtype=x_local.dtype, device=x_local.device)
        send[: x_local.shape[0]] = x_local
    out = torch.empty((world * per, nv), dtype=x_local.dtype, device=x_local.device)
    dist.all_gather_into_tensor(out, send.contiguous(), group=group)
    return out[:n_agents]


def reduce_safety_metrics(safety_ratio_local, vel_excess_local, acc_excess_local, group=None):
    """Mission-wide safety figures from the per-agent records of lscqp_safety_metrics_device on every rank
    (reference src/multi_sync_simulator.cpp:512-514, 560-572: running min of the safety ratio, running max of the excess
    ratios): tensors (n_local,), (n_local, 3), (n_local, 3) -> (safety_ratio_agent, vel_excess_ratio[3], acc_excess_ratio[3])
    on every rank.  One MIN and one MAX all-reduce of 1 + 6 doubles; an empty shard contributes +inf / 0."""
    dev, dt = safety_ratio_local.device, safety_ratio_local.dtype
    mn = safety_ratio_local.min().reshape(1) if safety_ratio_local.numel() else torch.full((1,), float("inf"), dtype=dt, device=dev)
    if vel_excess_local.numel():
        mx = torch.cat([vel_excess_local.max(dim=0).values, acc_excess_local.max(dim=0).values])
    else:
        mx = torch.zeros(6, dtype=dt, device=dev)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    return mn[0], mx[:3], mx[3:]


def exchange_plan_buffers(buffers, n_total, group=None):
    """The exchange step of a sharded replan chain (lscqp_plan with first_agent / n_agents = shard_range(n_total, world, rank)): every
    rank holds the buffers of ALL agents -- previous plans, states, goal points: what the next replan's obstacle prediction, range
    filter and constraint generation read of the other agents (reference src/multi_sync_simulator.cpp:305-352, broadcastMsgs) --
    and has just rewritten its own block.  `buffers`: 1-D contiguous tensors of n_total * k elements each (api.Plan.tensor(which) for
    PLAN_PLAN, PLAN_STATE, PLAN_GOAL: zero-copy views of the plan's device buffers), updated IN PLACE: one broadcast per owner and
    buffer (blocks may be ragged; k doubles per agent, ~1 KB for a plan: latency bound whatever the collective).  The single-process
    analogue behind the C ABI is lscqp_plan_group_step."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    work = []
    for buf in buffers:
        assert buf.dim() == 1 and buf.is_contiguous() and buf.numel() % n_total == 0
        per = buf.numel() // n_total
        for r in range(world):
            lo, hi = shard_range(n_total, world, r)
            if hi > lo:
                src = r if group is None else dist.get_global_rank(group, r)
                work.append(dist.broadcast(buf[lo * per:hi * per], src=src, group=group, async_op=True))
    for w in work:
        w.wait()
