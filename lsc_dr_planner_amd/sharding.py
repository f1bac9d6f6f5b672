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
