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
          which files them as ghost vertices (phl_add_vertices) and answers with the ORDER it wants
          their rows in: first the vertices it has itself (their rows are added to its own sums),
          then the ghosts in the order of its vertex buffer -- nearest to the cut first -- so that
          a ghost row is received straight into its place.
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

Ghost rows cost blur work, so each blur axis computes only the rows something later reads
(RowBand._plan_blur_rows: the sets are derived from the neighbour tables themselves, going back
from "slice reads the band's own vertices" through the axes; with the ghosts ordered by distance
from the cut every set is the own rows plus a PREFIX of each neighbour's ghosts).

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
    """One rank's share.  Build phases: build_outbox -> build_inbox -> order_outbox -> order_inbox; then per call
    splat_outbox -> finish.  Tensors live on the engine's device."""

    def __init__(self, feat, rank, world, engine_factory, device, table=None):
        """table: "reference" -- the band is CUT OUT OF the whole image's reference-table lattice (engines that offer
        ``whole_image`` / ``sub_lattice``: every rank builds the whole lattice once, keeps its band and the ghosts, and
        needs no build-time exchange at all: vertex ids of the whole lattice are common knowledge) -- or "clean": one
        defect-free lattice per band from its own pixels, ghosts exchanged by key.  Default: "reference" where the engine
        can (PHL_ROWTILE_TABLE=clean overrides)."""
        H, W, d = feat.shape
        self.rank, self.world, self.W, self.d = rank, world, W, d
        if table is None:
            table = "reference" if (hasattr(engine_factory, "whole_image") and os.environ.get("PHL_ROWTILE_TABLE", "reference") != "clean") else "clean"
        self.table = table
        cuts = band_rows(H, world)
        self.row0, self.row1 = cuts[rank], cuts[rank + 1]
        self.own_rows = self.row1 - self.row0
        self.n_local = self.own_rows * W
        k, sign, g, a_k, b_k = band_axis(feat)
        self.S = int(math.ceil((2 * a_k + b_k) / g - 1e-9)) + 1
        self.axis, self.reach, self._sign = k, a_k + b_k, sign
        y = sign * feat[..., k]
        span = [(float(y[r0:r1].min()), float(y[r0:r1].max())) for r0, r1 in zip(cuts[:-1], cuts[1:])]
        for r in range(world - 2):
            if span[r + 2][0] - span[r][1] <= 2 * a_k + b_k:
                raise ValueError(f"row bands ({min(r1 - r0 for r0, r1 in zip(cuts[:-1], cuts[1:]))} rows) are shorter than the "
                                 f"lattice support ({self.S} rows): use fewer ranks")
        self._span = span[rank]
        self.device = device
        own = np.ascontiguousarray(feat[self.row0:self.row1].reshape(-1, d), dtype=np.float32)
        t0 = time.time()
        self.sides, self._send_keys, self._ready = {}, {}, False
        reach = (a_k + b_k) * _REACH_SCALE + 1e-3 * (1.0 + max(abs(span[0][0]), abs(span[-1][1])))   # + fp32 elevate slack
        if table == "reference":
            self._cut_from_whole(feat, cuts, span, sign, k, reach, engine_factory, own)
            self._t_build = time.time() - t0
            return
        self.eng = engine_factory(torch.from_numpy(own).to(device))
        self.M_own = int(self.eng.M)          # before any ghost is filed
        keys = self.eng.keys()
        ypos = sign * vertex_coordinate(keys, d, k)
        for peer in (rank - 1, rank + 1):
            if 0 <= peer < world:
                v = np.nonzero((ypos >= span[peer][0] - reach) & (ypos <= span[peer][1] + reach))[0]
                # keys() numbers vertices in first-touch order; the vertex buffers' rows may be ordered differently
                rows = self.eng.vertex_rows().cpu().numpy()[v] if hasattr(self.eng, "vertex_rows") else v
                self.sides[peer] = dict(send_idx=torch.from_numpy(np.asarray(rows).astype(np.int64)).to(device))
                self._send_keys[peer] = torch.from_numpy(np.ascontiguousarray(keys[v]))
        self._t_build = time.time() - t0
        self.peers = sorted(self.sides)

    @property
    def needs_exchange(self):
        """False once everything the filter phases need is in place (a band cut from the whole lattice never exchanges
        anything at build time)."""
        return not self._ready

    def _cut_from_whole(self, feat, cuts, span, sign, k, reach, factory, own_feat):
        """table = "reference": see __init__.  Both sides of a cut derive the same lists from the whole lattice:
        what rank s sends rank t = s's vertices within a_k + b_k of t's span, first those t has itself (ascending vertex
        id), then t's ghosts, nearest to t's band first (ties by vertex id) -- the order of t's ghost rows."""
        H, W, d = feat.shape
        rank, world, dev = self.rank, self.world, self.device
        whole = factory.whole_image(torch.from_numpy(np.ascontiguousarray(feat.reshape(-1, d), dtype=np.float32)).to(dev))
        keys = whole.keys()
        y = sign * vertex_coordinate(keys, d, k)
        near = {r: (y >= span[r][0] - reach) & (y <= span[r][1] + reach) for r in (rank - 1, rank, rank + 1) if 0 <= r < world}
        dist = {r: np.maximum(0.0, np.maximum(span[r][0] - y, y - span[r][1])) for r in near}
        has = {r: whole.vertices_of_pixels(cuts[r] * W, cuts[r + 1] * W) for r in near}
        own = np.nonzero(has[rank])[0]
        peers = [p for p in (rank - 1, rank + 1) if 0 <= p < world]

        def wanted_by(t, s):
            """ids rank s sends rank t, in t's order; number of shared ones"""
            cand = has[s] & near[t]
            shared = np.nonzero(cand & has[t])[0]
            ghosts = np.nonzero(cand & ~has[t])[0]
            ghosts = ghosts[np.lexsort((ghosts, dist[t][ghosts]))]
            return shared, ghosts

        sel, plan = [own], {}
        for p in peers:
            shared_in, ghosts_in = wanted_by(rank, p)
            shared_out, ghosts_out = wanted_by(p, rank)
            plan[p] = (shared_in, ghosts_in, shared_out, ghosts_out)
            sel.append(ghosts_in)
        sel = np.concatenate(sel).astype(np.int32)
        self.eng = whole.sub_lattice(cuts[rank] * W, cuts[rank + 1] * W, sel, len(own), torch.from_numpy(own_feat).to(dev))
        self.M_whole = int(whole.M)
        if hasattr(whole, "close"):
            whole.close()
        self.M_own = len(own)
        pos = np.full(len(keys), -1, np.int64)
        pos[sel] = np.arange(len(sel))                          # whole-lattice vertex -> the band's first-touch id
        rows = self.eng.vertex_rows().cpu().numpy().astype(np.int64) if hasattr(self.eng, "vertex_rows") else np.arange(len(sel))
        g0 = len(own)
        for p in peers:
            shared_in, ghosts_in, shared_out, ghosts_out = plan[p]
            ghost_rows = rows[pos[ghosts_in]]
            assert np.array_equal(ghost_rows, g0 + np.arange(len(ghosts_in))), "ghost rows must follow the own rows in the given order"
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(dev)
            self.sides[p] = dict(send_idx=t(rows[pos[np.concatenate([shared_out, ghosts_out])]]), send_shared=len(shared_out),
                                 n_shared=len(shared_in), n_ghost=len(ghosts_in), ghost_row0=g0, shared_map=t(rows[pos[shared_in]]),
                                 map_idx=t(np.concatenate([rows[pos[shared_in]], ghost_rows])))
            g0 += len(ghosts_in)
        self.peers = sorted(self.sides)
        self._send_keys = None
        self._finalize()

    # -- build phases ---------------------------------------------------------------------------
    def build_outbox(self):
        """{peer: int16 [K, d] keys of my vertices within a_k + b_k of the peer's band}"""
        return dict(self._send_keys)

    def build_inbox(self, inbox):
        """File the neighbours' vertices: ghosts of one neighbour get consecutive rows, nearest to my band first."""
        t0 = time.time()
        lo, hi = self._span
        self._order_out = {}
        for peer in sorted(inbox):
            keys = inbox[peer].cpu().numpy().reshape(-1, self.d)
            y = self._sign * vertex_coordinate(keys, self.d, self.axis)
            dist = np.maximum(0.0, np.maximum(lo - y, y - hi))
            order = np.argsort(dist, kind="stable")
            before = int(self.eng.M)
            ids = np.asarray(self.eng.add_vertices(keys[order])).astype(np.int64)
            ghost = ids >= before
            # rows of new vertices follow the order they were handed in: ghost rows of this peer = [before, before + ng)
            assert np.array_equal(ids[ghost], before + np.arange(int(ghost.sum()))), "engine must append new vertices in call order"
            perm = np.concatenate([order[~ghost], order[ghost]]).astype(np.int32)
            ns = int((~ghost).sum())
            s = self.sides[peer]
            s["n_shared"], s["n_ghost"], s["ghost_row0"] = ns, int(ghost.sum()), before
            s["shared_map"] = torch.from_numpy(ids[~ghost]).to(self.device)
            s["map_idx"] = torch.from_numpy(np.concatenate([ids[~ghost], ids[ghost]])).to(self.device)
            self._order_out[peer] = torch.from_numpy(np.concatenate([[ns], perm]).astype(np.int32))
        self._t_build += time.time() - t0
        self._send_keys = None

    def order_outbox(self):
        """{peer: int32 [1 + K]}: how many of the K rows the peer sends me are vertices I have myself, then the order I
        want all K in (indices into the key list it sent: shared first, then ghosts in the order of my rows)."""
        return dict(self._order_out)

    def order_inbox(self, inbox):
        t0 = time.time()
        for peer in sorted(inbox):
            msg = inbox[peer].cpu().numpy().astype(np.int64)
            s = self.sides[peer]
            s["send_shared"] = int(msg[0])
            s["send_idx"] = s["send_idx"][torch.from_numpy(msg[1:]).to(self.device)]
        self._finalize()
        self._t_build += time.time() - t0

    def _finalize(self):
        # packed forms for the engine's fused row kernels: one gather for all peers' send rows, one scatter-add for
        # all received rows of shared vertices (only if no vertex receives from both sides: no atomics)
        if self.peers:
            self._send_all = torch.cat([self.sides[p]["send_idx"] for p in self.peers])
            self._map_all = torch.cat([self.sides[p]["map_idx"] for p in self.peers])
            self._shared_all = torch.cat([self.sides[p]["shared_map"] for p in self.peers])
            self._map_disjoint = int(torch.unique(self._map_all).numel()) == int(self._map_all.numel())
            self._send_unique = int(torch.unique(self._send_all).numel()) == int(self._send_all.numel())
            self._send_rng, self._recv_rng, self._shared_rng, so, ro, sh = {}, {}, {}, 0, 0, 0
            for p in self.peers:
                s = self.sides[p]
                ks, kr = int(s["send_idx"].numel()), int(s["map_idx"].numel())
                self._send_rng[p], self._recv_rng[p] = (so, so + ks), (ro, ro + kr)
                self._shared_rng[p] = (sh, sh + s["n_shared"])
                so, ro, sh = so + ks, ro + kr, sh + s["n_shared"]
        self._plan_blur_rows()
        self._ready = True

    def _plan_blur_rows(self):
        """Which rows each blur axis has to produce.  Slice reads the band's own vertices (rows [0, M_own)); axis j's
        output at row v is read by axis j+1 at v and at v's two neighbours along j+1.  Going back from the last axis gives,
        per axis, the exact set of rows whose output is ever used; ghosts are ordered by distance from the band, so each
        set is the own rows plus a prefix of every neighbour's ghost rows (the prefix up to the farthest needed one)."""
        eng = self.eng
        self.blur_rows = None
        if not (hasattr(eng, "set_blur_rows") and self.peers and int(eng.M) > self.M_own):
            return
        if os.environ.get("PHL_ROWTILE_BLUR_ROWS", "1") == "0":
            return
        M, d = int(eng.M), self.d
        nb = np.asarray(eng.neighbors()).astype(np.int64)            # [d+1, M, 2], first-touch ids, -1 absent
        rows = eng.vertex_rows().cpu().numpy().astype(np.int64)      # first-touch id -> row
        nbr = np.empty_like(nb)
        nbr[:, rows, :] = np.where(nb >= 0, rows[np.clip(nb, 0, None)], -1)     # row -> neighbour rows
        need = np.zeros(M, bool)
        need[:self.M_own] = True
        ranges = np.zeros((d + 1, 3, 2), np.int64)
        for axis in range(d, -1, -1):
            ranges[axis, 0] = (0, self.M_own)
            for k, p in enumerate(self.peers):
                s = self.sides[p]
                g0, ng = s["ghost_row0"], s["n_ghost"]
                hit = np.nonzero(need[g0:g0 + ng])[0]
                ranges[axis, 1 + k] = (g0, g0 + (int(hit[-1]) + 1 if hit.size else 0))
            if len(self.peers) == 1:
                ranges[axis, 2] = (ranges[axis, 1, 1], ranges[axis, 1, 1])
            nn = nbr[axis, need].ravel()
            need[nn[nn >= 0]] = True
        eng.set_blur_rows(ranges)
        self.blur_rows = ranges

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

    def place(self, vert, inbox):
        """Received rows -> the vertex buffer: rows of vertices this band has itself are added, ghost rows (no local
        contribution: the splat left zeros there) are copied into their consecutive rows.  Peers in ascending order."""
        fused = hasattr(self.eng, "scatter_add_rows") and vert.shape[1] % 4 == 0
        for peer in sorted(inbox):
            s, rows = self.sides[peer], inbox[peer]
            ns, ng, g0 = s["n_shared"], s["n_ghost"], s["ghost_row0"]
            if ng:
                vert[g0:g0 + ng].copy_(rows[ns:ns + ng])
            if ns:
                if fused:
                    self.eng.scatter_add_rows(vert, s["shared_map"], rows[:ns])
                else:
                    vert.index_add_(0, s["shared_map"], rows[:ns].to(vert.device))
        return vert

    def finish(self, vert, inbox, out=None, scratch=None, sub=None):
        """inbox: {peer: rows in the order asked for}; scratch: optional second [M, C] buffer for the blur."""
        self.place(vert, inbox)
        return self.blur_slice(vert, out=out, scratch=scratch, sub=sub)

    def blur_slice(self, vert, out=None, scratch=None, sub=None):
        """sub: subtract these rows from the result inside the slice (the ``- U`` of LatticeGaussian, gaussian_matrix.py:303)."""
        vert = self.eng.blur(vert) if scratch is None else self.eng.blur(vert, scratch)
        if sub is None:
            return self.eng.slice(vert) if out is None else self.eng.slice(vert, out=out)
        if hasattr(self.eng, "gather_rows"):           # engines with the phl.Lattice stage surface fuse it
            return self.eng.slice(vert, sub=sub, out=out)
        res = self.eng.slice(vert) - sub
        if out is not None:
            out.copy_(res)
            return out
        return res

    @property
    def M(self):
        return self.eng.M

    def recv_rows(self, peer):
        return int(self.sides[peer]["map_idx"].numel())


class RowTileFilter:
    """torch.distributed driver: one RowBand per rank, exchanges by batched isend/irecv.

    Schedules, best first:
      * edge chunks first (one channel group; engines with the chunk splat): the pixel chunks that feed this rank's
        boundary vertices are splatted first, the interior chunks right behind them; a SIDE stream completes the
        boundary rows into the send buffer (one kernel: sums + pack) as soon as the edge chunks are done and hands them
        to the exchange, which then runs under the interior chunks and the completion of all other rows.  Ghost rows
        are received straight into the vertex buffer, shared rows are added, blur (restricted rows) and slice follow.
        Payloads stay in HBM under RCCL; under gloo they are staged through pinned host buffers -- same schedule.
      * channel groups pipelined (device payloads): the boundary rows of group g travel while group g+1 is splatted.
      * plain: any engine, any backend."""

    def __init__(self, feat, L, rank, world, device, dist, engine_factory=None, groups=None, table=None):
        if engine_factory is None:
            import phl
            engine_factory = phl.Lattice
        self.dist, self.L, self.device = dist, L, device
        # P2P payloads must live where the backend can reach them
        self.comm_device = device if dist.get_backend() == "nccl" else torch.device("cpu")
        t0 = time.time()
        self.band = band = RowBand(feat, rank, world, engine_factory, device, table=table)
        if band.needs_exchange:              # bands built from their own pixels: ghosts by key, then the order of their rows
            out = band.build_outbox()
            counts = self._exchange({p: torch.tensor([k.shape[0]], dtype=torch.int64) for p, k in out.items()},
                                    {p: ((1,), torch.int64) for p in out})
            inbox = self._exchange(out, {p: ((int(c.item()), band.d), torch.int16) for p, c in counts.items()})
            band.build_inbox(inbox)
            band.order_inbox(self._exchange(band.order_outbox(), {p: ((1 + int(k.shape[0]),), torch.int32) for p, k in out.items()}))
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        self.build_ms = (time.time() - t0) * 1e3
        self.row0, self.own_rows, self.n_local = band.row0, band.own_rows, band.n_local
        self._direct = self.comm_device == device                        # payloads stay on the engine's device
        self._rows_engine = hasattr(band.eng, "gather_rows") and device.type == "cuda"
        want_edge = (os.environ.get("PHL_ROWTILE_EDGE_FIRST", "1") != "0" and self._rows_engine and hasattr(band.eng, "splat_part")
                     and L % 4 == 0 and bool(band.sides) and band._map_disjoint)
        if groups is None:
            groups = int(os.environ.get("PHL_ROWTILE_GROUPS", "0"))
            if not groups:
                groups = 1 if (want_edge or not (world > 1 and L % 8 == 0 and L >= 64)) else 2
        self.groups = [(g * L // groups, (g + 1) * L // groups) for g in range(groups)]
        self._fused = self._rows_engine and self._direct and (not band.sides or band._map_disjoint) and all((c1 - c0) % 4 == 0 for c0, c1 in self.groups)
        self._plan, self._edge_first = None, False
        # default: edge first on one queue (the boundary rows are certainly ready after a third of the splat, fewest
        # cross-stream dependencies, least host time); with nothing on the wire the three forms are within 3 % of each other
        # on one GPU (0.33-0.34 ms for an interior band of 8), autotune() picks by measurement with the real exchange
        self._mode, self._tuned = os.environ.get("PHL_ROWTILE_MODE", "edge, one queue"), None
        self._stub_exchange = False      # timing probes only: run the step without its point-to-point exchange
        self._vert = self._scratch = self._sbuf = self._ops = self._rshared = None
        if hasattr(band.eng, "reserve"):
            band.eng.reserve(max(c1 - c0 for c0, c1 in self.groups))
        if want_edge and groups == 1:
            self._make_edge_plan()
        if self._fused or self._edge_first:
            self._make_buffers()
        if not (self._fused or self._edge_first):
            total = sum(band.recv_rows(p) for p in band.sides)
            self._rpack = [torch.empty((total, c1 - c0), dtype=torch.float32, device=self.comm_device) for c0, c1 in self.groups]
            self._rbuf = [{p: pack[slice(*band.recv_range(p))] for p in band.sides} for pack in self._rpack]

    # ---- steady-state buffers and op lists (nothing is allocated or rebuilt per call) -------------------------
    def _make_buffers(self):
        band, dev, dist = self.band, self.device, self.dist
        M = band.M
        pin = dict(pin_memory=True) if (not self._direct and dev.type == "cuda") else {}
        self._vert = [torch.empty((M, c1 - c0), dtype=torch.float32, device=dev) for c0, c1 in self.groups]
        self._scratch = [torch.empty((M, c1 - c0), dtype=torch.float32, device=dev) for c0, c1 in self.groups[:1]]
        self._scratch += [self._scratch[0] if (c1 - c0) == self._scratch[0].shape[1] else
                          torch.empty((M, c1 - c0), dtype=torch.float32, device=dev) for c0, c1 in self.groups[1:]]
        self._sbuf = [torch.empty((band.send_rows(), c1 - c0), dtype=torch.float32, device=dev) for c0, c1 in self.groups]
        n_shared = sum(band.sides[p]["n_shared"] for p in band.peers)
        self._rshared = [torch.empty((n_shared, c1 - c0), dtype=torch.float32, device=dev) for c0, c1 in self.groups]
        if not self._direct:             # host staging (gloo): pinned mirrors of the send buffer and of both receive parts
            n_ghost = sum(band.sides[p]["n_ghost"] for p in band.peers)
            self._sbuf_h = [torch.empty(tuple(b.shape), dtype=torch.float32, **pin) for b in self._sbuf]
            self._rshared_h = [torch.empty(tuple(b.shape), dtype=torch.float32, **pin) for b in self._rshared]
            self._rghost_h = [torch.empty((n_ghost, c1 - c0), dtype=torch.float32, **pin) for c0, c1 in self.groups]
        self._ops = []
        for gi in range(len(self.groups)):
            ops, go = [], 0
            for p in band.peers:
                s = band.sides[p]
                a, b = band._send_rng[p]
                sa, sb = band._shared_rng[p]
                ns_out = s["send_shared"]
                snd = self._sbuf[gi] if self._direct else self._sbuf_h[gi]
                rsh = self._rshared[gi] if self._direct else self._rshared_h[gi]
                rgh = (self._vert[gi][s["ghost_row0"]:s["ghost_row0"] + s["n_ghost"]] if self._direct
                       else self._rghost_h[gi][go:go + s["n_ghost"]])
                go += s["n_ghost"]
                # two messages per direction: the rows the receiver adds to its own, the rows it takes as they are
                for t, op in ((snd[a:a + ns_out], dist.isend), (snd[a + ns_out:b], dist.isend), (rsh[sa:sb], dist.irecv), (rgh, dist.irecv)):
                    if t.shape[0]:
                        ops.append(dist.P2POp(op, t, p))
            self._ops.append(ops)
        if self.device.type == "cuda":
            self._side = torch.cuda.Stream(device=dev, priority=-1)       # high priority: the boundary goes first
            self._ev_start, self._ev_edge, self._ev_packed = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()

    def _make_edge_plan(self):
        """Edge-first schedule (one channel group): the chunks that feed this rank's boundary vertices are
        splatted first and those rows completed, so that they travel while the interior chunks are splatted --
        the exchange hides behind ~2/3 of the splat instead of behind a second channel group (splitting the
        channels costs 10 % of the step in small-launch overhead)."""
        eng, band = self.band.eng, self.band
        if not eng.tile_stats(self.L)["staged_splat"]:
            return
        send = band._send_all
        mask = eng.chunks_touching(send)
        if mask.all() or not mask.any():
            return
        dev = band.device
        order = torch.argsort(send)                       # the rows the edge part completes, ascending ...
        send_rows = send[order]
        if band._send_unique:
            pack_pos = order.to(torch.int32)              # ... and where each goes in the send buffer
        else:
            send_rows, pack_pos = torch.unique(send_rows), None        # (a row wanted by both neighbours: packed by a gather)
        is_other = torch.ones(band.M_own, dtype=torch.bool, device=dev)
        is_other[send_rows] = False                       # ghost rows belong to neither part: they are received
        self._plan = dict(
            edge=torch.from_numpy(np.nonzero(mask)[0].astype(np.int32)).to(dev),
            interior=torch.from_numpy(np.nonzero(~mask)[0].astype(np.int32)).to(dev),
            send_rows=send_rows.to(torch.int32).contiguous(), pack_pos=pack_pos,
            other_rows=torch.nonzero(is_other).flatten().to(torch.int32),
            none=torch.empty(0, dtype=torch.int32, device=dev),
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

    def filter(self, src, out=None, subtract_input=False):
        """src: this rank's rows, [own_rows*W, L] on the engine device -> same shape (written to ``out`` if given).
        subtract_input: ``filter(src) - src`` with the subtraction fused into the slice (LatticeGaussian's product)."""
        dist = self.dist
        if out is None:
            out = torch.empty((self.band.n_local, self.L), dtype=torch.float32, device=self.band.device)
        self._sub = src if subtract_input else None
        if self._edge_first and self._mode != "whole":
            return self._filter_edge_serial(src, out) if self._mode == "edge, one queue" else self._filter_edge_first(src, out)
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
            self.band.finish(vert, inbox, out=out[:, c0:c1], sub=None if self._sub is None else self._sub[:, c0:c1])
        return out

    def _filter_edge_serial(self, src, out):
        """Edge first on ONE queue (round 3's schedule): edge chunks, their rows summed and packed, the exchange handed over,
        then the interior chunks behind them on the same stream.  The two chunk launches meet at a kernel boundary (one
        interior band of 8: +30 us against the two-queue form with nothing on the wire), but the boundary rows are certainly
        ready after the first third of the splat, whatever the hardware's arbitration between two running kernels does."""
        band, eng, pl, dist = self.band, self.band.eng, self._plan, self.dist
        vert, sbuf, none = self._vert[0], self._sbuf[0], pl["none"]
        main = torch.cuda.current_stream(self.device)
        eng.splat_part(src, vert, pl["partial"], pl["edge"], pl["send_rows"], pack_pos=pl["pack_pos"], pack=sbuf if pl["pack_pos"] is not None else None)
        if pl["pack_pos"] is None:
            eng.gather_rows(vert, band._send_all, out=sbuf)
        reqs = []
        if self._direct:
            if self._ops[0] and not self._stub_exchange:
                reqs = dist.batch_isend_irecv(self._ops[0])
        else:
            self._sbuf_h[0].copy_(sbuf, non_blocking=True)
            self._ev_packed.record(main)
        eng.splat_part(src, vert, pl["partial"], pl["interior"], pl["other_rows"])
        if not self._direct:
            self._ev_packed.synchronize()
            if self._ops[0] and not self._stub_exchange:
                reqs = dist.batch_isend_irecv(self._ops[0])
        for req in reqs:
            req.wait()
        self._receive_staged(vert)
        if band._shared_all.numel():
            eng.scatter_add_rows(vert, band._shared_all, self._rshared[0])
        band.blur_slice(vert, out=out, scratch=self._scratch[0], sub=self._sub)
        return out

    def _receive_staged(self, vert):
        """host-staged payloads: ghost rows to their place, shared rows next to it (nothing to do with device payloads)"""
        if self._direct:
            return
        band, go = self.band, 0
        for p in band.peers:
            s = band.sides[p]
            if s["n_ghost"]:
                vert[s["ghost_row0"]:s["ghost_row0"] + s["n_ghost"]].copy_(self._rghost_h[0][go:go + s["n_ghost"]], non_blocking=True)
            go += s["n_ghost"]
        self._rshared[0].copy_(self._rshared_h[0], non_blocking=True)

    def _filter_edge_first(self, src, out):
        band, eng, pl, dist = self.band, self.band.eng, self._plan, self.dist
        vert, sbuf, none = self._vert[0], self._sbuf[0], pl["none"]
        main, side = torch.cuda.current_stream(self.device), self._side
        # Two queues.  The HIGH-PRIORITY side stream carries the boundary: edge chunks -> their rows summed and packed by one
        # kernel -> the exchange.  The caller's stream carries the bulk: interior chunks -> all other rows.  The two chunk
        # launches are independent (each writes its own partial rows / sole rows), so the interior workgroups fill the
        # slots the edge workgroups leave -- no kernel boundary between them, the splat costs what one launch over all
        # chunks costs -- while the edge part, first in line, is done after about a third of it.
        self._ev_start.record(main)
        reqs = []
        with torch.cuda.stream(side):
            side.wait_event(self._ev_start)     # (the previous step's slice has finished with `vert`)
            eng.splat_part(src, vert, pl["partial"], pl["edge"], none)
            self._ev_edge.record(side)
            eng.splat_part(src, vert, pl["partial"], none, pl["send_rows"], pack_pos=pl["pack_pos"], pack=sbuf if pl["pack_pos"] is not None else None)
            if pl["pack_pos"] is None:
                eng.gather_rows(vert, band._send_all, out=sbuf)
            if self._direct:
                if self._ops[0] and not self._stub_exchange:
                    reqs = dist.batch_isend_irecv(self._ops[0])               # ... and on their way (ordered behind `side`)
            else:
                self._sbuf_h[0].copy_(sbuf, non_blocking=True)
                self._ev_packed.record(side)
        eng.splat_part(src, vert, pl["partial"], pl["interior"], none)        # interior chunks, beside the edge part
        main.wait_event(self._ev_edge)          # rows shared between an edge and an interior chunk need both
        eng.splat_part(src, vert, pl["partial"], none, pl["other_rows"])      # all other rows of this band, under the exchange
        if not self._direct:
            self._ev_packed.synchronize()       # the host waits for the boundary rows only; everything else is queued
            if self._ops[0] and not self._stub_exchange:
                reqs = dist.batch_isend_irecv(self._ops[0])
        for req in reqs:
            req.wait()
        main.wait_stream(side)
        self._receive_staged(vert)
        if band._shared_all.numel():
            eng.scatter_add_rows(vert, band._shared_all, self._rshared[0])
        band.blur_slice(vert, out=out, scratch=self._scratch[0], sub=self._sub)
        return out

    def _filter_fused(self, src, out):
        """Channel groups on preallocated buffers and persistent op lists (device payloads)."""
        dist, band = self.dist, self.band
        reqs = []
        for gi, (c0, c1) in enumerate(self.groups):
            band.splat_outbox(src[:, c0:c1], vert=self._vert[gi], sendbuf=self._sbuf[gi])
            reqs.append(dist.batch_isend_irecv(self._ops[gi]) if (self._ops[gi] and not self._stub_exchange) else [])
        for gi, (c0, c1) in enumerate(self.groups):
            for req in reqs[gi]:
                req.wait()
            if band.sides and band._shared_all.numel():
                band.eng.scatter_add_rows(self._vert[gi], band._shared_all, self._rshared[gi])
            band.blur_slice(self._vert[gi], out=out[:, c0:c1], scratch=self._scratch[gi], sub=None if self._sub is None else self._sub[:, c0:c1])
        return out

    MODES = ("edge, two queues", "edge, one queue", "whole")

    def autotune(self, src, out, reps=6):
        """Pick the step's schedule by MEASUREMENT, with the real exchange (collective: every rank must call it).  How much
        of the exchange a schedule hides depends on things this code cannot know in advance -- how the hardware arbitrates
        two concurrently running kernels, which hardware queue the communication library's stream shares, how long the wire
        takes -- so the three forms that give bit-identical results are each timed over `reps` steps (barrier, MAX over the
        ranks) and the fastest is kept:
          "edge, two queues"   edge chunks on a high-priority side stream beside the interior chunks (fastest with nothing on
                               the wire: no kernel boundary inside the splat);
          "edge, one queue"    edge chunks, then the interior chunks behind them (boundary rows certainly ready after a third
                               of the splat);
          "whole"              the whole splat, then the exchange (nothing hidden, no split at all).
        Returns {mode: ms}; the choice is in describe()."""
        if not self._edge_first:
            return None
        dist, dev = self.dist, self.band.device
        modes = [m for m in self.MODES if m != "whole" or self._fused]
        res = {}
        for m in modes:
            self._mode = m
            for _ in range(2):
                self.filter(src, out=out)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                self.filter(src, out=out)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            dt = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64, device=dev if self._direct else "cpu")
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            res[m] = round(float(dt.item()), 4)
        self._mode = min(res, key=res.get)          # (the same on every rank: the times were reduced over the ranks)
        self._tuned = res
        return res

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
                if self._ops is not None:
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
        blur = None
        if getattr(b, "blur_rows", None) is not None:
            blur = [int((b.blur_rows[a, :, 1] - b.blur_rows[a, :, 0]).sum()) for a in range(b.d + 1)]
        return {"rowtile": {"table": b.table, "rows_per_rank": b.own_rows, "strip_rows": b.S, "M_local_plus_ghosts": int(b.M), "M_own": int(b.M_own),
                            "boundary_vertices_recv": rows,
                            "ghost_rows_received_in_place": {str(p): b.sides[p]["n_ghost"] for p in b.sides},
                            "blur_rows_per_axis": blur,
                            "channel_groups": len(self.groups),
                            "schedule": ((("edge chunks first, exchange under the interior splat" if self._mode != "whole" else "whole splat, then the exchange")
                                          + ("" if self._direct else " (payloads staged through pinned host memory)"))
                                         if self._edge_first else ("channel groups pipelined" if self._fused else "plain")),
                            "mode": self._mode if self._edge_first else None, "autotune_ms": self._tuned,
                            "edge_chunks": int(self._plan["edge"].numel()) if self._plan else None,
                            "exchange_bytes_per_step_per_rank": int(sum(rows.values()) * self.L * 4 * 2)}}


class RowBandGaussian:
    """The W of ``mean_field_infer(E_0, W, Mu)`` (crf/crf_module.py:41-53) for ONE RANK of a row-band run: ``W @ U`` is
    LatticeGaussian's product ``filter(U, ref) - U`` (crf/gaussian_matrix.py:292-303) on this rank's rows of the image, the
    boundary vertices exchanged with the neighbouring ranks inside.  Everything else of a mean-field iteration -- the
    compatibility product, the softmax -- is per pixel, so every rank runs ``mean_field_infer`` on its own rows of E_0 with
    this operator and the ranks' results stacked are the whole image's.  Collective: all ranks must call ``@`` the same
    number of times."""

    def __init__(self, feat, L, rank, world, device, dist, **kw):
        self.job = RowTileFilter(feat, L, rank, world, device, dist, **kw)
        self.row0, self.own_rows, self.n_local = self.job.row0, self.job.own_rows, self.job.n_local

    def __matmul__(self, U):
        assert U.shape == (self.n_local, self.job.L), "Incompatible shapes {}, and {}".format(tuple(U.shape), (self.n_local, self.job.L))
        return self.job.filter(U, subtract_input=True)

    def rows(self, full):
        """this rank's rows of a whole-image [H*W, .] tensor"""
        W = self.job.band.W
        return full[self.row0 * W:(self.row0 + self.own_rows) * W]


def simulate(feat, src_full, world, engine_factory, device, table=None):
    """Play all `world` ranks in this process (loopback exchange).  Returns the filtered image
    [H*W, L] assembled from the bands.  For tests on a single GPU / CPU."""
    H, W, d = feat.shape
    bands = [RowBand(feat, r, world, engine_factory, device, table=table) for r in range(world)]
    if any(b.needs_exchange for b in bands):
        out = [b.build_outbox() for b in bands]
        for r, b in enumerate(bands):
            b.build_inbox({p: out[p][r] for p in b.sides})
        order = [b.order_outbox() for b in bands]
        for r, b in enumerate(bands):
            b.order_inbox({p: order[p][r] for p in b.sides})
    outs = [b.splat_outbox(src_full[b.row0 * W:b.row1 * W]) for b in bands]
    res = []
    for r, b in enumerate(bands):
        res.append(b.finish(outs[r][0], {p: outs[p][1][r] for p in b.sides}))
    return torch.cat(res, 0), bands
