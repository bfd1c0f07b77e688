"""A process group made of THREADS (test infrastructure): the surface of torch.distributed that sml_amd.dist uses,
for N ranks that are N Python threads of one process sharing ONE GPU.  It lets the -m gpu suite drive the real HIP
library under world_size 2 -- unequal local batches, item-gradient exchange, theta all-reduce, replicated item tables --
on a single-GPU box: every rank has its own HipEngine (own sml_ctx, own stream, own tables); collectives are host
rendezvous (a barrier) around plain tensor copies.  Collectives are issued in the same order by every rank, as with
any process group."""
import threading

import torch


class ReduceOp(object):
    SUM, MIN, MAX = "sum", "min", "max"


class ThreadGroup(object):
    ReduceOp = ReduceOp

    def __init__(self, world):
        self.world = world
        self._barrier = threading.Barrier(world)
        self._slots = [None] * world
        self._local = threading.local()

    # ---- identity
    def bind(self, rank):
        self._local.rank = rank

    def get_world_size(self, group=None):
        return self.world

    def get_rank(self, group=None):
        return self._local.rank

    def get_backend(self, group=None):
        return "threads"

    def is_initialized(self):
        return True

    # ---- rendezvous: every rank deposits, all wait, every rank reads, all wait again before the slots are reused
    def _exchange(self, value):
        if isinstance(value, torch.Tensor) and value.is_cuda:
            torch.cuda.current_stream(value.device).synchronize()      # the depositor's producers have finished
        self._slots[self.get_rank()] = value
        self._barrier.wait()
        got = list(self._slots)
        return got

    def _done(self):
        self._barrier.wait()

    def barrier(self, group=None):
        self._barrier.wait()

    def all_reduce(self, t, op=ReduceOp.SUM, group=None):
        got = self._exchange(t.clone())
        acc = got[0].clone()
        for x in got[1:]:                       # rank order: every rank forms the identical result
            acc = acc + x if op == ReduceOp.SUM else (torch.minimum(acc, x) if op == ReduceOp.MIN else torch.maximum(acc, x))
        t.copy_(acc)
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
        self._done()

    def all_gather(self, outs, src, group=None):
        got = self._exchange(src.clone())
        for o, x in zip(outs, got):
            o.copy_(x)
        if src.is_cuda:
            torch.cuda.current_stream(src.device).synchronize()
        self._done()

    def all_gather_into_tensor(self, dst, src, group=None):
        self.all_gather(list(dst.view(self.world, -1).unbind(0)), src.reshape(-1))

    def broadcast(self, t, src=0, group=None):
        got = self._exchange(t.clone())
        t.copy_(got[src])
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
        self._done()

    def all_gather_object(self, out, obj, group=None):
        got = self._exchange(obj)
        out[:] = got
        self._done()

    def broadcast_object_list(self, box, src=0, group=None):
        got = self._exchange(list(box))
        box[:] = got[src]
        self._done()


def run_ranks(world, fn, *args, streams=None):
    """fn(rank, group, *args) on `world` threads; returns the list of results (re-raises the first failure).
    streams: one torch stream per rank (e.g. CU-masked streams over disjoint compute units, so that kernels of one rank
    that POLL for another rank's kernels can never keep those off the chip); default: a fresh stream per rank."""
    group = ThreadGroup(world)
    out, err = [None] * world, [None] * world

    def body(rank):
        group.bind(rank)
        try:
            if torch.cuda.is_available():
                with torch.cuda.stream(streams[rank] if streams is not None else torch.cuda.Stream()):
                    out[rank] = fn(rank, group, *args)
                    torch.cuda.synchronize()
            else:
                out[rank] = fn(rank, group, *args)
        except BaseException as e:              # noqa: BLE001 -- reported by the caller; and free the peers
            err[rank] = e
            if not group._barrier.broken:       # the FIRST failure is the cause; what follows is ranks finding the barrier broken
                import sys
                import traceback
                sys.stderr.write("rank %d failed first:\n" % rank)
                traceback.print_exc()
            group._barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out
