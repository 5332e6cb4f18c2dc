"""Row-band sharding of ONE image over several GPUs (north star: "images shard by row-tile
across the 8 GPUs of one node with a RCCL halo exchange over xGMI").

The reference has no counterpart (single process, single thread); the only parallelism on its
path is one image per worker (crf/gaussian_matrix.py:370-377), which here is simply one Lattice
per GPU and needs no communication.

Decomposition (one process per GPU, rank r owns image rows [row0, row1)):

  build   every rank builds the lattice of ITS OWN pixels.  Splat is a sum over pixels, so a
          vertex near a cut gets contributions from two bands, and blur reaches vertices that
          only the neighbouring band creates.  Each rank therefore sends the KEYS of the
          vertices touched by its pixels within S rows of a cut to the rank across that cut,
          which files them as ghost vertices (phl_add_vertices) and remembers the index map.
  filter  splat own pixels -> exchange the partial sums of those boundary vertices with the
          <= 2 neighbouring ranks (point-to-point, RCCL over xGMI) -> add -> blur -> slice own
          pixels.  One exchange per filter call; message = (#boundary vertices) x L floats.

Strip depth S.  Let f be a feature that grows by at least g per image row (the "y" feature).
A pixel's simplex vertices lie within the simplex diameter a = sqrt(d+1)*sqrt(6)/4 feature
units of it; the d+1 blur steps displace by at most b = sqrt(6(d+1))/2 in total along any
direction; so the vertices whose COMPLETE splat sums a band needs lie within a+b of its own
pixels, and the foreign pixels contributing to them within 2a+b = sqrt(6(d+1)) (6.0 units for
d=5).  S = ceil(sqrt(6(d+1)) / g) rows.  Vertices deeper in the strip arrive with incomplete
sums; by the same bound they cannot influence the band's own pixels.  Results equal the
single-lattice filter up to fp32 summation order (own partial + neighbour partial instead of
one pixel-ordered sum): ~1e-7 relative, asserted at 1e-4 in tests.

The class is engine-agnostic (anything with the phl.Lattice stage surface) and phase-structured
(outbox / inbox), so the same code runs under torch.distributed (NCCL on GPUs, gloo on CPU in the
tests, where the engine is the CPU oracle) and under an in-process loopback that plays all ranks
on one GPU (tests/test_gpu_rowtile.py).
"""
import math
import os
import time

import numpy as np
import torch


def strip_rows(feat, d=None):
    """S for an [H, W, d] feature image: ceil(sqrt(6(d+1)) / g) + 1, g = the largest guaranteed
    per-row increment of any feature."""
    H, W, dd = feat.shape
    d = dd if d is None else d
    if H < 2:
        raise ValueError("row bands need at least 2 image rows")
    inc = np.diff(feat.astype(np.float64), axis=0)            # [H-1, W, d]
    lo = inc.min(axis=(0, 1))
    hi = inc.max(axis=(0, 1))
    g = float(np.max(np.where(lo > 0, lo, np.where(hi < 0, -hi, 0.0))))
    if g <= 0:
        raise ValueError("row-band sharding needs a feature that is strictly monotone in the row index "
                         "(e.g. y/sigma); these features have none")
    return int(math.ceil(math.sqrt(6.0 * (d + 1)) / g)) + 1


def band_rows(H, world):
    return [(H * r) // world for r in range(world + 1)]


class RowBand:
    """One rank's share.  Phases: build_outbox -> build_inbox, then per call splat_outbox ->
    finish.  Tensors live on the engine's device."""

    def __init__(self, feat, rank, world, engine_factory, device):
        H, W, d = feat.shape
        self.rank, self.world, self.W, self.d = rank, world, W, d
        cuts = band_rows(H, world)
        self.row0, self.row1 = cuts[rank], cuts[rank + 1]
        self.own_rows = self.row1 - self.row0
        self.n_local = self.own_rows * W
        self.S = strip_rows(feat)
        if world > 1 and min(b - a for a, b in zip(cuts[:-1], cuts[1:])) < self.S:
            raise ValueError(f"row bands ({min(b - a for a, b in zip(cuts[:-1], cuts[1:]))} rows) are shorter than the "
                             f"lattice support ({self.S} rows): use fewer ranks")
        self.device = device
        own = np.ascontiguousarray(feat[self.row0:self.row1].reshape(-1, d), dtype=np.float32)
        t0 = time.time()
        self.eng = engine_factory(torch.from_numpy(own).to(device))
        self.sides = {}   # peer -> dict(send_idx, map_idx)
        vid, _ = self.eng.replay()
        keys = self.eng.keys()
        vid = vid.reshape(self.own_rows, W, d + 1)
        self._send_keys = {}
        for peer, rows in ((rank - 1, slice(0, self.S)), (rank + 1, slice(self.own_rows - self.S, self.own_rows))):
            if 0 <= peer < world:
                v = np.unique(vid[rows].ravel())
                self.sides[peer] = dict(send_idx=torch.from_numpy(v.astype(np.int64)).to(device))
                self._send_keys[peer] = torch.from_numpy(np.ascontiguousarray(keys[v]))
        self._t_build = time.time() - t0

    # -- build phases ---------------------------------------------------------------------------
    def build_outbox(self):
        """{peer: int16 [K, d] keys of my vertices touched within S rows of the cut}"""
        return dict(self._send_keys)

    def build_inbox(self, inbox):
        t0 = time.time()
        for peer in sorted(inbox):
            ids = self.eng.add_vertices(inbox[peer].cpu().numpy())
            self.sides[peer]["map_idx"] = torch.from_numpy(ids.astype(np.int64)).to(self.device)
        self._t_build += time.time() - t0
        self._send_keys = None

    # -- filter phases --------------------------------------------------------------------------
    def splat_outbox(self, src):
        """src [n_local, C] (any channel subset, unit column stride) -> (vertex sums, {peer: rows to send})"""
        vert = self.eng.splat(src)
        return vert, {peer: vert.index_select(0, s["send_idx"]) for peer, s in self.sides.items()}

    def finish(self, vert, inbox, out=None):
        for peer in sorted(inbox):
            vert.index_add_(0, self.sides[peer]["map_idx"], inbox[peer])   # distinct rows: deterministic
        vert = self.eng.blur(vert)
        return self.eng.slice(vert) if out is None else self.eng.slice(vert, out=out)

    @property
    def M(self):
        return self.eng.M

    def recv_rows(self, peer):
        return int(self.sides[peer]["map_idx"].numel())


class RowTileFilter:
    """torch.distributed driver: one RowBand per rank, exchanges by batched isend/irecv.

    The filter is channel-wise, so a call is pipelined over `groups` channel groups: the boundary
    rows of group g travel over xGMI (RCCL runs on its own stream) while group g+1 is being
    splatted, and group g is blurred/sliced while group g+1's rows are still in flight."""

    def __init__(self, feat, L, rank, world, device, dist, engine_factory=None, groups=None):
        if engine_factory is None:
            import phl
            engine_factory = phl.Lattice
        self.dist, self.L, self.device = dist, L, device
        # P2P payloads must live where the backend can reach them
        self.comm_device = device if dist.get_backend() == "nccl" else torch.device("cpu")
        t0 = time.time()
        self.band = RowBand(feat, rank, world, engine_factory, device)
        out = self.band.build_outbox()
        counts = self._exchange({p: torch.tensor([k.shape[0]], dtype=torch.int64) for p, k in out.items()},
                                {p: ((1,), torch.int64) for p in out})
        inbox = self._exchange(out, {p: ((int(c.item()), self.band.d), torch.int16) for p, c in counts.items()})
        self.band.build_inbox(inbox)
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        self.build_ms = (time.time() - t0) * 1e3
        self.row0, self.own_rows, self.n_local = self.band.row0, self.band.own_rows, self.band.n_local
        if groups is None:
            groups = int(os.environ.get("PHL_ROWTILE_GROUPS", "0")) or (2 if (world > 1 and L % 8 == 0 and L >= 64) else 1)
        self.groups = [(g * L // groups, (g + 1) * L // groups) for g in range(groups)]
        self._rbuf = [{p: torch.empty((self.band.recv_rows(p), c1 - c0), dtype=torch.float32, device=self.comm_device)
                       for p in self.band.sides} for c0, c1 in self.groups]
        if hasattr(self.band.eng, "reserve"):
            self.band.eng.reserve(max(c1 - c0 for c0, c1 in self.groups))

    @property
    def M(self):
        return self.band.M

    def _exchange(self, outbox, recv_spec):
        """Send outbox[peer], receive a tensor of recv_spec[peer] = (shape, dtype) from each peer."""
        dist = self.dist
        ops, recv, keep = [], {}, []
        for peer in sorted(recv_spec):
            shape, dtype = recv_spec[peer]
            snd = outbox[peer].to(self.comm_device).contiguous()
            recv[peer] = torch.empty(shape, dtype=dtype, device=self.comm_device)
            rcv = recv[peer]
            if dtype == torch.int16:          # RCCL has no 16-bit integer type: ship the bytes
                snd, rcv = snd.view(torch.uint8), rcv.view(torch.uint8)
            keep.append(snd)
            ops.append(dist.P2POp(dist.isend, snd, peer))
            ops.append(dist.P2POp(dist.irecv, rcv, peer))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return recv

    def filter(self, src):
        """src: this rank's rows, [own_rows*W, L] on the engine device -> same shape."""
        dist = self.dist
        out = torch.empty((self.band.n_local, self.L), dtype=torch.float32, device=self.band.device)
        pending = []
        for gi, (c0, c1) in enumerate(self.groups):
            vert, outbox = self.band.splat_outbox(src[:, c0:c1])
            ops, keep = [], []
            for peer in sorted(outbox):
                snd = outbox[peer] if outbox[peer].device == self.comm_device else outbox[peer].to(self.comm_device)
                keep.append(snd)
                ops.append(dist.P2POp(dist.isend, snd.contiguous(), peer))
                ops.append(dist.P2POp(dist.irecv, self._rbuf[gi][peer], peer))
            reqs = dist.batch_isend_irecv(ops) if ops else []
            pending.append((vert, reqs, keep))
        for gi, (c0, c1) in enumerate(self.groups):
            vert, reqs, _ = pending[gi]
            for req in reqs:
                req.wait()
            inbox = {p: (b if b.device == self.band.device else b.to(self.band.device)) for p, b in self._rbuf[gi].items()}
            self.band.finish(vert, inbox, out=out[:, c0:c1])
        return out

    def describe(self):
        b = self.band
        rows = {str(p): b.recv_rows(p) for p in b.sides}
        return {"rowtile": {"rows_per_rank": b.own_rows, "strip_rows": b.S, "M_local_plus_ghosts": int(b.M),
                            "boundary_vertices_recv": rows,
                            "channel_groups": len(self.groups),
                            "exchange_bytes_per_step_per_rank": int(sum(rows.values()) * self.L * 4 * 2)}}


def simulate(feat, src_full, world, engine_factory, device):
    """Play all `world` ranks in this process (loopback exchange).  Returns the filtered image
    [H*W, L] assembled from the bands.  For tests on a single GPU / CPU."""
    H, W, d = feat.shape
    bands = [RowBand(feat, r, world, engine_factory, device) for r in range(world)]
    out = [b.build_outbox() for b in bands]
    for r, b in enumerate(bands):
        b.build_inbox({p: out[p][r] for p in b.sides})
    outs = [b.splat_outbox(src_full[b.row0 * W:b.row1 * W]) for b in bands]
    res = [b.finish(outs[r][0], {p: outs[p][1][r] for p in b.sides}) for r, b in enumerate(bands)]
    return torch.cat(res, 0), bands
