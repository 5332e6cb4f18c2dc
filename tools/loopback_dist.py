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


class WireDist(LoopbackDist):
    """One rank alone with an exchange that TAKES TIME: a batch is enqueued on a communication stream behind the caller's
    current stream (RCCL's ordering) and fills the receive buffers there with a kernel that has the footprint of a
    point-to-point transfer -- `workgroups` workgroups copying from a zero buffer, `repeat` times over (phl_debug_slow_copy):
    long-running, next to no HBM bandwidth, a few wave slots.  The numbers received are meaningless, the timing is not: how
    much of a given wire time does the step's schedule hide?  (band_time.py variants wire<workgroups>x<repeat>.)"""

    class _Req:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)
            return True

    def __init__(self, world, workgroups, repeat):
        super().__init__(world)
        import ctypes

        import phl

        self.lib = phl.load_library()
        self.lib.phl_debug_slow_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        self.c = ctypes
        import os

        # HIP maps streams onto a few hardware queues in creation order: dummy streams in front shift the one the exchange gets
        self._dummies = [torch.cuda.Stream() for _ in range(int(os.environ.get("WIRE_DUMMY_STREAMS", "0")))]
        self.comm = torch.cuda.Stream(priority=int(os.environ.get("WIRE_COMM_PRIORITY", "0")))
        self.workgroups, self.repeat = workgroups, repeat
        self.zero = None
        self.wire_ms = None

    def _copies(self, ops):
        for o in ops:
            if o.op == "irecv" and o.tensor.numel():
                if self.zero is None or self.zero.numel() < o.tensor.numel():
                    self.zero = torch.zeros(o.tensor.numel(), dtype=torch.float32, device=o.tensor.device)
                assert o.tensor.is_contiguous() and o.tensor.numel() % 4 == 0
                rc = self.lib.phl_debug_slow_copy(self.c.c_void_p(self.zero.data_ptr()), self.c.c_void_p(o.tensor.data_ptr()), o.tensor.numel(),
                                                  self.workgroups, self.repeat, self.c.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0

    def batch_isend_irecv(self, ops):
        if self.wire_ms is None:            # the exchange alone, once
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(self.comm):
                self._copies(ops)
                torch.cuda.synchronize()
                e0.record(self.comm)
                for _ in range(5):
                    self._copies(ops)
                e1.record(self.comm)
            torch.cuda.synchronize()
            self.wire_ms = e0.elapsed_time(e1) / 5
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.comm.wait_event(ev)
        with torch.cuda.stream(self.comm):
            self._copies(ops)
            done = torch.cuda.Event()
            done.record(self.comm)
        return [self._Req(done)]
