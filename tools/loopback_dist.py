"""In-process stand-in for torch.distributed with device-resident payloads (what RCCL gives the row-band driver): every
rank is a thread, isend / irecv meet in FIFO queues and copy GPU tensor to GPU tensor.  For the one-GPU timing tools
(band_time.py, band_prof.py); tests/test_gpu_rowtile.py keeps its own copy."""
import queue
import threading

import torch


class LoopbackDist:
    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Req:
        def wait(self):
            return True

    def __init__(self, world, backend="nccl"):
        self.world, self.backend = world, backend
        self.q = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
        self.local = threading.local()
        self.isend, self.irecv = "isend", "irecv"
        self._bar = threading.Barrier(world)

    def get_backend(self):
        return self.backend

    def barrier(self):
        self._bar.wait()

    def batch_isend_irecv(self, ops):
        me = self.local.rank
        for o in ops:
            if o.op == "isend":
                if o.tensor.is_cuda:
                    torch.cuda.synchronize()
                self.q[(me, o.peer)].put(o.tensor.clone())
        for o in ops:
            if o.op == "irecv":
                got = self.q[(o.peer, me)].get(timeout=300)
                if o.tensor.dtype == torch.uint8 or got.dtype == torch.uint8:
                    o.tensor.view(torch.uint8).copy_(got.view(torch.uint8))
                else:
                    o.tensor.copy_(got)
        return [self._Req() for _ in ops]


def build_jobs(feat, L, world, device, want_ranks=None, backend="nccl", **kw):
    """RowTileFilter of every rank (threads; the key / order exchanges of the build are collective among neighbours);
    returns {rank: job} for want_ranks (default: all)."""
    from phl import rowtile

    fake = LoopbackDist(world, backend)
    jobs, errs = {}, []
    import os

    import phl

    if (want_ranks is not None and kw.get("table", None) != "clean" and os.environ.get("PHL_ROWTILE_TABLE", "reference") != "clean"
            and hasattr(phl.Lattice, "whole_image")):
        # bands cut out of the whole image's lattice need no build-time exchange: build only the ranks asked for, one after
        # the other (as separate processes would: one whole-image build each)
        for r in want_ranks:
            fake.local.rank = r
            jobs[r] = rowtile.RowTileFilter(feat, L, r, world, device, fake, **kw)
        return jobs, fake

    def run(r):
        try:
            fake.local.rank = r
            jobs[r] = rowtile.RowTileFilter(feat, L, r, world, device, fake, **kw)
        except Exception:      # noqa: BLE001
            import traceback

            errs.append((r, traceback.format_exc()))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise RuntimeError(str(errs))
    keep = set(range(world) if want_ranks is None else want_ranks)
    return {r: j for r, j in jobs.items() if r in keep}, fake
