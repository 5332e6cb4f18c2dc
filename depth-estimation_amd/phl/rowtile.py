"""Row-band sharding of ONE image over several GPUs (north star: "images shard by row-tile
across the 8 GPUs of one node with a RCCL halo exchange over xGMI").

The reference has no counterpart (single process, single thread); the only parallelism on its
path is one image per worker (crf/gaussian_matrix.py:370-377), which here is simply one Lattice
per GPU and needs no communication.

Decomposition (one process per GPU, rank r owns image rows [row0, row1)):

  build   every rank builds the lattice of ITS OWN pixels.  Splat is a sum over pixels, so a
          vertex near a cut gets contributions from two bands, and blur reaches vertices that
          only the neighbouring band creates.  Each rank therefore sends the KEYS of the
          vertices the rank across a cut can need (selected by lattice coordinate, below) to that rank,
          which files them as ghost vertices (phl_add_vertices) and remembers the index map.
  filter  splat own pixels -> exchange the partial sums of those boundary vertices with the
          <= 2 neighbouring ranks (point-to-point, RCCL over xGMI) -> add -> blur -> slice own
          pixels.  One exchange per filter call; message = (#boundary vertices) x L floats.

Which vertices travel.  Feature k is elevated along the unit vector u_k = (1,..,1 [k+1 ones],
-(k+1), 0,..)/sqrt((k+1)(k+2)) of the lattice hyperplane, scaled by alpha = (d+1)*sqrt(2/3)
(permutohedral.h:354-384), so a vertex with integer key K sits at feature coordinate
y(K) = K.u_k / alpha -- exact, no geometry assumed.  Along that coordinate
  a_k  = how far a simplex vertex can lie from a pixel inside the simplex
       = max over simplex edges w (entries m x(d+1-m), -(d+1-m) x m, any order) of |w.u_k| / alpha,
  b_k  = the total displacement of the d+1 blur steps (step j moves by (1,..,1)-(d+1)e_j, whose
         u_k component is -(d+1)u_k[j])  = sqrt(3/2) * |u_k|_1
(d=5, k=1: a = 1.0, b = 2.0 feature units; the direction-free bounds are 1.5 and 3.0).
A band whose pixels span [lo, hi] in feature k reads, at slice time, vertices within a_k of that
interval, and their blurred values depend on splat sums within a further b_k.  So a rank sends
to the rank across a cut exactly its vertices with y(K) in [lo' - a_k - b_k, hi' + a_k + b_k]
(lo', hi' = the RECEIVER's span), which the receiver files as ghosts; nothing deeper is needed
and nothing deeper is sent.  Bands two apart must not share such vertices: their spans have to
be separated by more than 2a_k + b_k (checked at build; S = ceil((2a_k+b_k)/g)+1 rows is the
same bound in image rows for a feature growing by g per row).  Results equal the
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


def lattice_reach(d, k):
    """(a_k, b_k) of the module header, in units of feature k."""
    u = np.zeros(d + 1)
    u[:k + 1] = 1.0
    u[k + 1] = -(k + 1)
    u /= math.sqrt((k + 1) * (k + 2))
    alpha = (d + 1) * math.sqrt(2.0 / 3.0)
    us = np.sort(u)
    a = 0.0
    for m in range(1, d + 1):
        w = np.sort(np.array([m] * (d + 1 - m) + [-(d + 1 - m)] * m, dtype=np.float64))
        a = max(a, abs(float(w @ us)), abs(float(w @ us[::-1])))      # rearrangement inequality: extremes over orders
    return a / alpha, math.sqrt(1.5) * float(np.abs(u).sum())


def vertex_coordinate(keys, d, k):
    """Feature-k coordinate of lattice vertices given their int16 keys [M, d] (last coordinate implied)."""
    K = keys.astype(np.float64)
    # K_full . u_k with K_full = (K, -sum K); u_k is zero beyond index k+1
    full_k1 = K[:, k + 1] if k + 1 < d else -K.sum(axis=1)
    return (K[:, :k + 1].sum(axis=1) - (k + 1) * full_k1) / math.sqrt((k + 1) * (k + 2)) / ((d + 1) * math.sqrt(2.0 / 3.0))


def band_axis(feat):
    """Pick the feature the bands are ordered along: (k, sign, g, support) with sign*feat[..., k]
    growing by at least g > 0 per image row and support = 2a_k + b_k; the k with the fewest
    support rows wins."""
    H, W, d = feat.shape
    if H < 2:
        raise ValueError("row bands need at least 2 image rows")
    inc = np.diff(feat.astype(np.float64), axis=0)            # [H-1, W, d]
    lo, hi = inc.min(axis=(0, 1)), inc.max(axis=(0, 1))
    best = None
    for k in range(d):
        g, sign = (lo[k], 1.0) if lo[k] > 0 else ((-hi[k], -1.0) if hi[k] < 0 else (0.0, 0.0))
        if g > 0:
            a, b = lattice_reach(d, k)
            rows = (2 * a + b) / g
            if best is None or rows < best[0]:
                best = (rows, k, sign, float(g), a, b)
    if best is None:
        raise ValueError("row-band sharding needs a feature that is strictly monotone in the row index "
                         "(e.g. y/sigma); these features have none")
    return best[1:]


def strip_rows(feat):
    """Image rows the lattice support spans: ceil((2a_k + b_k) / g) + 1 (bands must be at least this tall)."""
    k, sign, g, a, b = band_axis(np.asarray(feat))
    return int(math.ceil((2 * a + b) / g - 1e-9)) + 1


_REACH_SCALE = 1.0   # tests shrink this to show the bound is tight, never the product


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
        k, sign, g, a_k, b_k = band_axis(feat)
        self.S = int(math.ceil((2 * a_k + b_k) / g - 1e-9)) + 1
        self.axis, self.reach = k, a_k + b_k
        y = sign * feat[..., k]
        span = [(float(y[r0:r1].min()), float(y[r0:r1].max())) for r0, r1 in zip(cuts[:-1], cuts[1:])]
        for r in range(world - 2):
            if span[r + 2][0] - span[r][1] <= 2 * a_k + b_k:
                raise ValueError(f"row bands ({min(r1 - r0 for r0, r1 in zip(cuts[:-1], cuts[1:]))} rows) are shorter than the "
                                 f"lattice support ({self.S} rows): use fewer ranks")
        self.device = device
        own = np.ascontiguousarray(feat[self.row0:self.row1].reshape(-1, d), dtype=np.float32)
        t0 = time.time()
        self.eng = engine_factory(torch.from_numpy(own).to(device))
        self.sides = {}   # peer -> dict(send_idx, map_idx)
        keys = self.eng.keys()
        ypos = sign * vertex_coordinate(keys, d, k)
        reach = (a_k + b_k) * _REACH_SCALE + 1e-3 * (1.0 + max(abs(span[0][0]), abs(span[-1][1])))   # + fp32 elevate slack
        self._send_keys = {}
        for peer in (rank - 1, rank + 1):
            if 0 <= peer < world:
                v = np.nonzero((ypos >= span[peer][0] - reach) & (ypos <= span[peer][1] + reach))[0]
                # keys() numbers vertices in first-touch order; the vertex buffers' rows may be ordered differently
                rows = self.eng.vertex_rows().cpu().numpy()[v] if hasattr(self.eng, "vertex_rows") else v
                self.sides[peer] = dict(send_idx=torch.from_numpy(np.asarray(rows).astype(np.int64)).to(device))
                self._send_keys[peer] = torch.from_numpy(np.ascontiguousarray(keys[v]))
        self._t_build = time.time() - t0

    # -- build phases ---------------------------------------------------------------------------
    def build_outbox(self):
        """{peer: int16 [K, d] keys of my vertices within a_k + b_k of the peer's band}"""
        return dict(self._send_keys)

    def build_inbox(self, inbox):
        t0 = time.time()
        for peer in sorted(inbox):
            ids = self.eng.add_vertices(inbox[peer].cpu().numpy())
            self.sides[peer]["map_idx"] = torch.from_numpy(ids.astype(np.int64)).to(self.device)
        # packed forms for the engine's fused row kernels: one gather for all peers' send rows, one
        # scatter-add for all received rows (only if no vertex receives from both sides: no atomics)
        self.peers = sorted(self.sides)
        if self.peers:
            self._send_all = torch.cat([self.sides[p]["send_idx"] for p in self.peers])
            self._map_all = torch.cat([self.sides[p]["map_idx"] for p in self.peers])
            self._map_disjoint = int(torch.unique(self._map_all).numel()) == int(self._map_all.numel())
            self._send_rng, self._recv_rng, so, ro = {}, {}, 0, 0
            for p in self.peers:
                ks, kr = int(self.sides[p]["send_idx"].numel()), int(self.sides[p]["map_idx"].numel())
                self._send_rng[p], self._recv_rng[p] = (so, so + ks), (ro, ro + kr)
                so, ro = so + ks, ro + kr
        self._t_build += time.time() - t0
        self._send_keys = None

    # -- filter phases --------------------------------------------------------------------------
    def splat_outbox(self, src, vert=None, sendbuf=None):
        """src [n_local, C] (any channel subset, unit column stride) -> (vertex sums, {peer: rows to send}).
        vert / sendbuf: optional preallocated [M, C] / [rows to send, C] buffers (engines with the phl.Lattice
        stage surface write into them, so a steady-state call allocates nothing)."""
        engine_rows = hasattr(self.eng, "gather_rows")
        vert = self.eng.splat(src, out=vert) if (vert is not None and engine_rows) else self.eng.splat(src)
        if self.sides and engine_rows:
            if vert.shape[1] % 4 == 0:       # the engine's row kernels move 16-byte pieces
                buf = self.eng.gather_rows(vert, self._send_all, out=sendbuf)        # one launch for both neighbours
            else:
                buf = torch.index_select(vert, 0, self._send_all, out=sendbuf) if sendbuf is not None else vert.index_select(0, self._send_all)
            return vert, {p: buf[a:b] for p, (a, b) in self._send_rng.items()}
        return vert, {peer: vert.index_select(0, s["send_idx"]) for peer, s in self.sides.items()}

    def send_rows(self):
        return int(self._send_all.numel()) if self.sides else 0

    def recv_range(self, peer):
        """Row range of `peer`'s rows inside a packed receive buffer (peers in ascending order)."""
        return self._recv_rng[peer]

    def finish(self, vert, inbox, out=None, packed=None, scratch=None):
        """inbox: {peer: rows}; packed: the same rows as ONE [sum of rows, C] tensor in ascending peer order
        (lets the engine add them in a single launch); scratch: optional second [M, C] buffer for the blur."""
        fused = hasattr(self.eng, "scatter_add_rows") and vert.shape[1] % 4 == 0
        if fused and packed is not None and self.sides and self._map_disjoint:
            self.eng.scatter_add_rows(vert, self._map_all, packed)
        else:
            for peer in sorted(inbox):      # distinct rows per peer, peers in a fixed order: deterministic
                if fused:
                    self.eng.scatter_add_rows(vert, self.sides[peer]["map_idx"], inbox[peer])
                else:
                    vert.index_add_(0, self.sides[peer]["map_idx"], inbox[peer])
        vert = self.eng.blur(vert) if scratch is None else self.eng.blur(vert, scratch)
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
            groups = int(os.environ.get("PHL_ROWTILE_GROUPS", "0"))
            if not groups:
                # device-resident payloads (RCCL): one group with the edge-first schedule (_make_edge_plan); payloads
                # staged through host memory (gloo): two channel groups pipelined
                edge_first = (self.comm_device == device and os.environ.get("PHL_ROWTILE_EDGE_FIRST", "1") != "0"
                              and hasattr(engine_factory, "splat_part"))
                groups = 1 if (edge_first or not (world > 1 and L % 8 == 0 and L >= 64)) else 2
        self.groups = [(g * L // groups, (g + 1) * L // groups) for g in range(groups)]
        total = sum(self.band.recv_rows(p) for p in self.band.sides)
        self._rpack = [torch.empty((total, c1 - c0), dtype=torch.float32, device=self.comm_device) for c0, c1 in self.groups]
        self._rbuf = [{p: pack[slice(*self.band.recv_range(p))] for p in self.band.sides} for pack in self._rpack]
        # Steady state allocates nothing and rebuilds nothing: vertex buffers, blur scratch, send buffers and the
        # point-to-point op lists of every channel group exist once (the host side of a 0.35 ms band step must
        # not be on the critical path).
        self._fused = hasattr(self.band.eng, "gather_rows") and self.comm_device == self.band.device
        self._vert = self._scratch = self._sbuf = self._ops = None
        self._plan, self._edge_first = None, False
        self._stub_exchange = False      # timing probes only: run the step without its point-to-point exchange
        if self._fused:
            M = self.band.M
            self._vert = [torch.empty((M, c1 - c0), dtype=torch.float32, device=device) for c0, c1 in self.groups]
            self._scratch = [torch.empty((M, c1 - c0), dtype=torch.float32, device=device) for c0, c1 in self.groups[:1]]
            self._scratch += [self._scratch[0] if (c1 - c0) == self._scratch[0].shape[1] else
                              torch.empty((M, c1 - c0), dtype=torch.float32, device=device) for c0, c1 in self.groups[1:]]
            self._sbuf = [torch.empty((self.band.send_rows(), c1 - c0), dtype=torch.float32, device=device) for c0, c1 in self.groups]
            self._ops = []
            for gi in range(len(self.groups)):
                ops = []
                for peer in sorted(self.band.sides):
                    a, b = self.band._send_rng[peer]
                    ops.append(dist.P2POp(dist.isend, self._sbuf[gi][a:b], peer))
                    ops.append(dist.P2POp(dist.irecv, self._rbuf[gi][peer], peer))
                self._ops.append(ops)
        if hasattr(self.band.eng, "reserve"):
            self.band.eng.reserve(max(c1 - c0 for c0, c1 in self.groups))
        self._make_edge_plan()

    def _make_edge_plan(self):
        """Edge-first schedule (one channel group): the chunks that feed this rank's boundary vertices are
        splatted first and those rows completed, so that they travel while the interior chunks are splatted --
        the exchange hides behind ~3/4 of the splat instead of behind a second channel group (splitting the
        channels costs 10 % of the step in small-launch overhead)."""
        eng, band = self.band.eng, self.band
        want = os.environ.get("PHL_ROWTILE_EDGE_FIRST", "1") != "0"
        if not (want and self._fused and len(self.groups) == 1 and band.sides and hasattr(eng, "splat_part")
                and self.L % 4 == 0 and eng.tile_stats(self.L)["staged_splat"]):
            return
        send = band._send_all
        mask = eng.chunks_touching(send)
        if mask.all() or not mask.any():
            return
        dev = band.device
        is_send = torch.zeros(band.M, dtype=torch.bool, device=dev)
        is_send[send] = True
        self._plan = dict(
            edge=torch.from_numpy(np.nonzero(mask)[0].astype(np.int32)).to(dev),
            interior=torch.from_numpy(np.nonzero(~mask)[0].astype(np.int32)).to(dev),
            send_rows=torch.nonzero(is_send).flatten().to(torch.int32),
            other_rows=torch.nonzero(~is_send).flatten().to(torch.int32),
            partial=torch.empty((max(eng.partial_rows, 1), self.L), dtype=torch.float32, device=dev))
        self._edge_first = True

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

    def filter(self, src, out=None):
        """src: this rank's rows, [own_rows*W, L] on the engine device -> same shape (written to ``out`` if given)."""
        dist = self.dist
        if out is None:
            out = torch.empty((self.band.n_local, self.L), dtype=torch.float32, device=self.band.device)
        if self._fused:
            return self._filter_fused(src, out)
        pending = []
        for gi, (c0, c1) in enumerate(self.groups):
            vert, outbox = self.band.splat_outbox(src[:, c0:c1])
            ops, keep = [], []
            for peer in sorted(outbox):
                snd = outbox[peer] if outbox[peer].device == self.comm_device else outbox[peer].to(self.comm_device)
                keep.append(snd)
                ops.append(dist.P2POp(dist.isend, snd.contiguous(), peer))
                ops.append(dist.P2POp(dist.irecv, self._rbuf[gi][peer], peer))
            reqs = dist.batch_isend_irecv(ops) if (ops and not self._stub_exchange) else []
            pending.append((vert, reqs, keep))
        for gi, (c0, c1) in enumerate(self.groups):
            vert, reqs, _ = pending[gi]
            for req in reqs:
                req.wait()
            pack = self._rpack[gi] if self._rpack[gi].device == self.band.device else self._rpack[gi].to(self.band.device)
            inbox = {p: pack[slice(*self.band.recv_range(p))] for p in self.band.sides}
            self.band.finish(vert, inbox, out=out[:, c0:c1], packed=pack)
        return out

    def _filter_edge_first(self, src, out):
        dist, band, eng, pl = self.dist, self.band, self.band.eng, self._plan
        vert, sbuf = self._vert[0], self._sbuf[0]
        eng.splat_part(src, vert, pl["partial"], pl["edge"], pl["send_rows"])        # boundary rows complete
        eng.gather_rows(vert, band._send_all, out=sbuf)
        reqs = dist.batch_isend_irecv(self._ops[0]) if (self._ops[0] and not self._stub_exchange) else []   # ... and on their way
        eng.splat_part(src, vert, pl["partial"], pl["interior"], pl["other_rows"])   # the rest, under the exchange
        for req in reqs:
            req.wait()
        band.finish(vert, self._rbuf[0], out=out, packed=self._rpack[0], scratch=self._scratch[0])
        return out

    def _filter_fused(self, src, out):
        """Same schedule on preallocated buffers and persistent op lists (RCCL: payloads stay in HBM)."""
        if self._edge_first:
            return self._filter_edge_first(src, out)
        dist, band = self.dist, self.band
        reqs = []
        for gi, (c0, c1) in enumerate(self.groups):
            band.splat_outbox(src[:, c0:c1], vert=self._vert[gi], sendbuf=self._sbuf[gi])
            reqs.append(dist.batch_isend_irecv(self._ops[gi]) if (self._ops[gi] and not self._stub_exchange) else [])
        for gi, (c0, c1) in enumerate(self.groups):
            for req in reqs[gi]:
                req.wait()
            band.finish(self._vert[gi], self._rbuf[gi], out=out[:, c0:c1], packed=self._rpack[gi], scratch=self._scratch[gi])
        return out

    def exchange_probe(self, src, out, reps=10):
        """Where the exchange stands in a step, measured on this rank (wall clock around synchronised loops, so it
        reads the same under RCCL and under gloo's host staging):
          exchange_ms            one step's point-to-point exchange alone, nothing overlapping it;
          step_ms                the whole step;
          step_no_exchange_ms    the same step with the exchange left out (stale boundary rows: timing only);
          overlap_hidden_frac    1 - (step - step_no_exchange) / exchange: the share of the exchange that the
                                 schedule hides behind compute (1 = free, 0 = fully exposed).
        Every rank must call it (the exchange is collective among neighbours)."""
        dist, dev = self.dist, self.band.device

        def sync():
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            dist.barrier()

        def timed(fn):
            fn()
            sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / reps * 1e3
            dist.barrier()
            return dt

        def exchange_only():
            for gi in range(len(self.groups)):
                if self._fused:
                    ops = self._ops[gi]
                else:
                    ops = []
                    for peer in sorted(self.band.sides):
                        a, b = self.band._send_rng[peer]
                        snd = torch.zeros((b - a, self.groups[gi][1] - self.groups[gi][0]), dtype=torch.float32, device=self.comm_device)
                        ops.append(dist.P2POp(dist.isend, snd, peer))
                        ops.append(dist.P2POp(dist.irecv, self._rbuf[gi][peer], peer))
                for req in (dist.batch_isend_irecv(ops) if ops else []):
                    req.wait()

        step_ms = timed(lambda: self.filter(src, out=out))
        self._stub_exchange = True
        try:
            noex_ms = timed(lambda: self.filter(src, out=out))
        finally:
            self._stub_exchange = False
        ex_ms = timed(exchange_only)
        self.filter(src, out=out)        # leave `out` holding a real result
        hidden = 1.0 - max(0.0, step_ms - noex_ms) / ex_ms if ex_ms > 0 else 1.0
        return {"exchange_ms": round(ex_ms, 4), "step_ms": round(step_ms, 4), "step_no_exchange_ms": round(noex_ms, 4),
                "overlap_hidden_frac": round(max(0.0, min(1.0, hidden)), 3)}

    def describe(self):
        b = self.band
        rows = {str(p): b.recv_rows(p) for p in b.sides}
        return {"rowtile": {"rows_per_rank": b.own_rows, "strip_rows": b.S, "M_local_plus_ghosts": int(b.M),
                            "boundary_vertices_recv": rows,
                            "channel_groups": len(self.groups),
                            "schedule": ("edge chunks first, exchange under the interior splat" if getattr(self, "_edge_first", False)
                                         else "channel groups pipelined"),
                            "edge_chunks": int(self._plan["edge"].numel()) if getattr(self, "_plan", None) else None,
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
    res = []
    for r, b in enumerate(bands):
        inbox = {p: outs[p][1][r] for p in b.sides}
        packed = torch.cat([inbox[p] for p in sorted(inbox)]) if inbox else None
        res.append(b.finish(outs[r][0], inbox, packed=packed))
    return torch.cat(res, 0), bands
