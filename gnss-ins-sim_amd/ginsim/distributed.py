"""Multi-GPU side of the engine: Monte-Carlo runs shard embarrassingly (contiguous global run ranges, one
process per GPU); the only exchange is ONE all-reduce of the per-device statistics record.

The record (count, mean[9], M2[9], max|e|[9]) is placed in row `rank` of a zero (world x 28) matrix and
summed with a single all-reduce (RCCL over xGMI on GPUs; gloo in the CPU tests).  Every rank then holds
all per-device records and folds them with the library's Chan merge -- same result on every rank,
independent of reduction order, and better conditioned than summing raw sum / sum-of-squares.
Payload: world x 28 doubles (1.8 KB at 8 GPUs) -> latency-bound; link bandwidth is irrelevant.
"""
import numpy as np

from .engine import StatsResult

RECORD = 28


def shard(total_runs, world, rank):
    """Contiguous, balanced split of [0,total_runs): returns (first_global_run, runs_on_this_rank)."""
    base, extra = divmod(int(total_runs), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def allreduce_stats(part, group=None, device=None):
    """Merge per-rank StatsResult objects across `group` with one all-reduce(SUM); identity when not
    distributed.  `device` is where the exchange tensor lives (cuda:<local_rank> for nccl, cpu for gloo)."""
    if group is None:
        return part
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    table = torch.zeros((world, RECORD), dtype=torch.float64, device=device)
    table[rank] = torch.from_numpy(part.pack()).to(table.device)
    dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)
    return StatsResult.merge(table.cpu().numpy())


def allreduce_stats_begin(part, group=None, device=None):
    """Non-blocking form: issue the all-reduce and return a handle for allreduce_stats_end.  The caller can enqueue more
    GPU work in between; the collective completes whenever the device gets to it."""
    if group is None:
        return part
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    table = torch.zeros((world, RECORD), dtype=torch.float64, device=device)
    table[rank] = torch.from_numpy(part.pack()).to(table.device)
    work = dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return work, table


def allreduce_stats_end(handle):
    if isinstance(handle, StatsResult):
        return handle
    work, table = handle
    work.wait()
    return StatsResult.merge(table.cpu().numpy())


def init_abi_comm(ctx, group=None, device=None):
    """Give `ctx` the library's own RCCL communicator over the ranks of a torch.distributed group: rank 0 draws the unique id
    (ginsim_comm_unique_id), torch.distributed only carries those 128 bytes to the other ranks (bootstrap, outside any timed
    region); afterwards MonteCarloJob.stats_all_begin / _finish exchange the records on the context's stream without torch.
    Every rank first checks that it can reach librccl at all and the verdicts are reduced, so that either all ranks enter the
    collective ncclCommInitRank or all of them raise (a rank that raised alone would leave the others waiting).
    `device`: where the small verdict tensor lives (cuda:<local_rank> with backend nccl, cpu with gloo)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    uid, problem = None, None
    try:
        uid = ctx.comm_unique_id()              # local call: loads librccl; only rank 0's id is used
    except Exception as e:                      # noqa: BLE001
        problem = repr(e)
    verdict = torch.tensor([0.0 if problem else 1.0], dtype=torch.float64, device=device)
    dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=group)
    if float(verdict.item()) < 1.0:
        raise RuntimeError('librccl is not usable on every rank%s' % (': ' + problem if problem else ''))
    box = [uid if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    ctx.comm_init(world, rank, box[0])
    return world, rank


def stats_from_errors(e):
    """Packed record of a host array of end-point errors (runs,9) -- used by tests to fabricate partials."""
    e = np.asarray(e, dtype=np.float64)
    m = e.mean(0)
    return np.concatenate([[e.shape[0]], m, ((e - m) ** 2).sum(0), np.abs(e).max(0)])
