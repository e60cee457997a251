"""Multi-GPU side of the engine: Monte-Carlo runs shard embarrassingly (contiguous global run ranges, one
process per GPU); the only exchange is ONE all-reduce of the per-device statistics record.

The record (count, mean[9], M2[9], max|e|[9]) is placed in row `rank` of a zero (world x 28) matrix and
summed with a single all-reduce (RCCL over xGMI on GPUs; gloo in the CPU tests).  Every rank then holds
all per-device records and folds them with the library's Chan merge -- same result on every rank,
independent of reduction order, and better conditioned than summing raw sum / sum-of-squares.
Payload: world x 28 doubles (1.8 KB at 8 GPUs) -> latency-bound; link bandwidth is irrelevant.
"""
import os

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


def _all_agree(ok, group, device):
    """True iff `ok` holds on every rank of the group (one small all-reduce(MIN))."""
    import torch
    import torch.distributed as dist
    flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return float(flag.item()) >= 1.0


def init_abi_comm(ctx, group=None, device=None):
    """Give `ctx` the library's own RCCL communicator over the ranks of a torch.distributed group: rank 0 draws the unique id
    (ginsim_comm_unique_id -- rank 0 only: the call starts RCCL's bootstrap listener), torch.distributed only carries those
    128 bytes to the other ranks (bootstrap, outside any timed region); afterwards MonteCarloJob.stats_all_begin / _finish
    exchange the records on the context's stream without torch.

    No rank may be left alone inside a collective: (1) every rank probes librccl (ginsim_comm_probe: dlopen + dlsym only) and
    the verdicts are reduced, so that either all ranks go on or all raise; (2) ncclCommInitRank is collective -- once every rank
    has entered it fails or succeeds on all of them -- and its outcome is reduced as well, so a context that did get a
    communicator drops it again when another rank did not; (3) that call is bounded in time ($GINSIM_COMM_INIT_TIMEOUT, 90 s).
    `device`: where the small verdict tensors live (cuda:<local_rank> with backend nccl, cpu with gloo)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    problem = None
    if getattr(ctx, '_comm_abandoned', None) is not None:      # reduced with the probe verdicts: all ranks go on or all raise
        problem = 'an earlier communicator bootstrap on this context timed out and was abandoned'
    else:
        try:
            ctx.comm_probe()
        except Exception as e:                  # noqa: BLE001
            problem = repr(e)
    if not _all_agree(problem is None, group, device):
        raise RuntimeError('librccl is not usable on every rank%s' % (': ' + problem if problem else ''))
    uid = None
    if rank == 0:
        try:
            uid = ctx.comm_unique_id()
        except Exception as e:                  # noqa: BLE001
            problem = repr(e)
    box = [uid]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if box[0] is None:                          # rank 0 could not draw an id: every rank sees that and raises
        raise RuntimeError('ncclGetUniqueId failed on rank 0%s' % (': ' + problem if problem else ''))
    # (3) bounded in time: ncclCommInitRank has no time-out of its own, and a first run on new hardware cannot be debugged --
    # the call runs on a helper thread (ctypes releases the GIL) and a rank that is not back after `limit` seconds reports
    # failure like any other; its thread is abandoned (daemon), the context gets no communicator
    import threading
    limit = float(os.environ.get('GINSIM_COMM_INIT_TIMEOUT', '90'))
    done, state, lock = {}, {'gave_up': False}, threading.Lock()

    def work():
        try:
            ctx.comm_init(world, rank, box[0])
        except Exception as e:                  # noqa: BLE001
            with lock:
                done['err'] = repr(e)
            return
        with lock:                              # one lock decides who owns the outcome: either the main thread sees 'ok' ...
            late = state['gave_up']
            if not late:
                done['ok'] = True
        if late:                                # ... or the ranks agreed long ago that this communicator does not exist: drop it
            ctx.comm_destroy()
    t = threading.Thread(target=work, name='ginsim-comm-init', daemon=True)
    t.start()
    t.join(limit)
    with lock:
        timed_out = not done                    # neither 'ok' nor 'err': the thread is still inside ncclCommInitRank
        if timed_out:
            state['gave_up'] = True
    if timed_out:
        # the context is poisoned -- it must not be destroyed under the thread (Context.close() leaks it while the thread lives),
        # and a communicator that appears later is dropped by the thread itself and never used
        ctx._comm_abandoned = t
        problem = 'ncclCommInitRank did not return within %g s' % limit
    elif 'err' in done:
        problem = done['err']
    if not _all_agree(problem is None, group, device):
        if problem is None:
            ctx.comm_destroy()
        raise RuntimeError('ncclCommInitRank failed on some rank%s' % (': ' + problem if problem else ''))
    return world, rank


class StatsExchange(object):
    """The ONE exchange of the Monte-Carlo path for a driver (Sim, bench.py): every rank's end-point record to every rank.

    With backend nccl the library's own RCCL all-gather behind the C ABI (ginsim_end_stats_all_begin / _finish on the kernel
    stream) is used -- after the communicator came up on EVERY rank and the first record it merged equalled the torch.distributed
    all-reduce of the same record on every rank; otherwise (gloo in the CPU tests, librccl missing, a disagreement) the
    torch.distributed all-reduce, and `note` says why.  One object per context; it is kept on the context."""

    def __init__(self, ctx, group, device):
        import torch.distributed as dist
        self.ctx, self.group, self.device = ctx, group, device
        self.kind, self.note, self._checked = 'torch', None, False
        if group is None:
            self.kind = None
            return
        if dist.get_backend(group) != 'nccl':
            self.note = 'backend %s' % dist.get_backend(group)
            return
        try:
            if not getattr(ctx, 'comm_ranks', 0):
                init_abi_comm(ctx, group, device)
            self.kind = 'abi'
        except Exception as e:                  # noqa: BLE001 -- every rank raised together (init_abi_comm): all fall back
            self.note = 'abi exchange unavailable: %s' % (repr(e)[:160],)

    @staticmethod
    def of(ctx, group, device):
        ex = getattr(ctx, '_stats_exchange', None)
        if ex is None or ex.group is not group:
            ex = ctx._stats_exchange = StatsExchange(ctx, group, device)
        return ex

    def merge(self, job, algo, ned=False):
        """Merged StatsResult of all ranks for the end-point record `algo` of `job` (None: this rank holds no runs)."""
        import ctypes as C
        from ._lib import lib, check, Stats
        if self.kind is None:
            return job.stats(algo, ned=ned)
        local = StatsResult.zero() if job is None else job.stats(algo, ned=ned)
        if self.kind != 'abi':
            return allreduce_stats(local, self.group, self.device)
        ptr, runs = (None, 0) if job is None else (job._bufs[('endned_' if ned else 'end_') + algo].ptr, job.runs)
        check(lib.ginsim_end_stats_all_begin(self.ctx.handle, ptr, runs, 7))
        s = Stats()
        check(lib.ginsim_end_stats_all_finish(self.ctx.handle, 7, C.byref(s)))
        merged = StatsResult(s)
        if not self._checked:                   # once per context: the two exchanges must agree on every rank
            ref = allreduce_stats(local, self.group, self.device)
            same = merged.count == ref.count and np.allclose(merged.mean, ref.mean, rtol=1e-12, atol=1e-18) and \
                np.allclose(merged.m2, ref.m2, rtol=1e-12) and np.array_equal(merged.maxabs, ref.maxabs)
            if _all_agree(same, self.group, self.device):
                self._checked = True
            else:
                self.ctx.comm_destroy()
                self.kind, self.note = 'torch', 'abi exchange disagreed with the torch.distributed all-reduce on the first record'
                return ref
        return merged


def stats_from_errors(e):
    """Packed record of a host array of end-point errors (runs,9) -- used by tests to fabricate partials."""
    e = np.asarray(e, dtype=np.float64)
    m = e.mean(0)
    return np.concatenate([[e.shape[0]], m, ((e - m) ** 2).sum(0), np.abs(e).max(0)])
