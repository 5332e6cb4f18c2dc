"""GPU: row-band decomposition with the HIP engine, all ranks played on one GPU (loopback), against
the single-lattice HIP filter; checks ghost-vertex insertion (phl_add_vertices) on the device."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def rel(a, b):
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * np.abs(b).max())).max())


@pytest.mark.parametrize("world,H,W,L", [(2, 128, 96, 16), (4, 256, 64, 32), (8, 512, 48, 8)])
def test_bands_match_single_lattice_on_gpu(world, H, W, L):
    import phl
    from phl import rowtile
    from test_rowtile_cpu import make_image

    feat, src = make_image(H, W, L, sigma_xy=3.0)
    dev = torch.device("cuda")
    s = torch.from_numpy(src).to(dev)
    want = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev), reference_table=True).filter(s).cpu().numpy()
    got, bands = rowtile.simulate(feat, s, world, phl.Lattice, dev)
    assert rel(got.cpu().numpy(), want) <= RTOL
    for b in bands:
        assert b.M >= b.eng.M_local and (b.M > b.eng.M_local or world == 1)


@pytest.mark.parametrize("vd", [4, 8, 20, 64, 128, 256, 260])
def test_row_gather_and_scatter_add(vd):
    import phl

    rng = np.random.default_rng(vd)
    ref = (rng.random((500, 3), dtype=np.float32) * 3).astype(np.float32)
    L = phl.Lattice(torch.from_numpy(ref).cuda())
    vert = torch.from_numpy(rng.standard_normal((L.M, vd)).astype(np.float32)).cuda()
    idx = torch.from_numpy(rng.permutation(L.M)[:L.M // 3].astype(np.int64)).cuda()
    got = L.gather_rows(vert, idx)
    assert torch.equal(got, vert.index_select(0, idx))
    rows = torch.from_numpy(rng.standard_normal((idx.numel(), vd)).astype(np.float32)).cuda()
    want = vert.clone().index_add_(0, idx, rows)
    big = torch.cat([rows, rows], 1)[:, :vd]                    # a strided view as the receive buffer slice
    assert torch.equal(L.scatter_add_rows(vert.clone(), idx, rows), want)
    assert torch.equal(L.scatter_add_rows(vert.clone(), idx, big), want)
    assert L.gather_rows(vert, idx[:0]).shape == (0, vd)


def test_add_vertices_semantics():
    import phl
    from oracle import phl_oracle as po

    rng = np.random.default_rng(2)
    ref = (rng.random((3000, 3), dtype=np.float32) * 4).astype(np.float32)
    L = phl.Lattice(torch.from_numpy(ref).cuda())
    O = po.Oracle(ref)
    keys = L.keys()
    M0 = L.M
    q = np.concatenate([keys[10:20], keys[:3] + np.int16(100), keys[50:52]])       # existing, new, existing
    ids = L.add_vertices(q)                   # ROWS of the vertex buffers
    oids = O.add_vertices(q)                  # first-touch vertex ids
    rows = L.vertex_rows().cpu().numpy()
    assert np.array_equal(ids, rows[oids]) and L.M == O.M == M0 + 3 and L.M_local == M0
    assert np.array_equal(oids[:10], np.arange(10, 20)) and np.array_equal(ids[10:13], np.arange(M0, M0 + 3))
    assert np.array_equal(L.keys(), O.keys()) and np.array_equal(L.neighbors(), O.neighbors())
    src = rng.standard_normal((3000, 8)).astype(np.float32)
    a = L.filter(torch.from_numpy(src).cuda(), exact=True).cpu().numpy()
    assert np.array_equal(a.view(np.uint32), O.filter(src).view(np.uint32))
    v = L.splat(torch.from_numpy(src).cuda())
    assert v.shape[0] == M0 + 3 and float(v[M0:].abs().max()) == 0.0               # ghosts receive nothing locally


def _gpu_worker(rank, world, port, H, W, L, q, groups):
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "depth-estimation_amd"), root, os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    from phl import rowtile
    from test_rowtile_cpu import make_image

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    feat, src = make_image(H, W, L, sigma_xy=3.0)
    job = rowtile.RowTileFilter(feat, L, rank, world, dev, dist, groups=groups)   # HIP engine; payloads staged through gloo
    mine = torch.from_numpy(src[job.row0 * W:(job.row0 + job.own_rows) * W]).to(dev)
    out = job.filter(mine)
    out2 = job.filter(mine)
    assert torch.equal(out, out2)
    probe = job.exchange_probe(mine, torch.empty_like(out), reps=2)
    assert torch.equal(job.filter(mine), out), "the stubbed-exchange timing pass must leave no trace"
    tuned = job.autotune(mine, torch.empty_like(out), reps=2)          # collective; None for the plain path
    if tuned is not None:
        assert set(tuned) == {"edge, two queues", "edge, one queue"} and job._mode in tuned, tuned     # (host-staged: no "whole")
        assert torch.equal(job.filter(mine), out), "whatever schedule won, the result is the same"
    info = job.describe()
    info["probe"] = probe
    q.put((rank, out.cpu().numpy(), info))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("groups", [None, 2])
def test_two_processes_share_the_gpu_over_gloo(groups):
    """The torch.distributed driver (RowTileFilter) with the HIP engine in two real processes.
    One GPU box cannot host two RCCL ranks, so the payloads travel through gloo here; the rank
    logic, ghost import and index maps are the ones the RCCL run uses.  groups = None: the DEFAULT schedule -- edge
    chunks first, boundary rows completed and packed on a side stream, exchange under the interior chunks, ghost rows
    received in place -- with its payloads staged through pinned host memory; groups = 2: the plain path."""
    import socket

    import torch.multiprocessing as mp

    import phl
    from test_rowtile_cpu import make_image

    H, W, L, world = 96, 64, 8, 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, H, W, L, q, groups)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    feat, src = make_image(H, W, L, sigma_xy=3.0)
    want = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).cuda(), reference_table=True).filter(torch.from_numpy(src).cuda()).cpu().numpy()
    got = np.concatenate([r[1] for r in res], 0)
    assert rel(got, want) <= RTOL
    sched = res[0][2]["rowtile"]["schedule"]
    if groups is None:
        assert sched.startswith("edge chunks first") and "pinned host" in sched, sched
        assert res[0][2]["rowtile"]["blur_rows_per_axis"] is not None
    else:
        assert sched == "plain", sched


class _LoopbackDist:
    """In-process stand-in for torch.distributed with device-resident payloads (what RCCL gives the driver):
    every rank is a thread, isend/irecv meet in queues and copy GPU tensor to GPU tensor.  Lets the
    preallocated-buffer / persistent-op path of RowTileFilter -- the one the RCCL run takes -- execute on a
    one-GPU box."""

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Req:
        def wait(self):
            return True

    def __init__(self, world):
        import queue
        import threading

        self.world = world
        self.q = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
        self.local = threading.local()
        self.isend, self.irecv = "isend", "irecv"

    def get_backend(self):
        return "nccl"

    def batch_isend_irecv(self, ops):
        me = self.local.rank
        for o in ops:
            if o.op == "isend":
                torch.cuda.synchronize()
                self.q[(me, o.peer)].put(o.tensor.clone())
        for o in ops:
            if o.op == "irecv":
                got = self.q[(o.peer, me)].get(timeout=120)
                if o.tensor.dtype == torch.uint8 or got.dtype == torch.uint8:
                    o.tensor.view(torch.uint8).copy_(got.view(torch.uint8))
                else:
                    o.tensor.copy_(got)
        return [self._Req() for _ in ops]


@pytest.mark.parametrize("world,groups", [(2, 1), (3, 2), (4, 2)])
def test_preallocated_rccl_shaped_path_in_process(world, groups):
    """RowTileFilter's steady-state path (device payloads, preallocated vertex / send / receive buffers,
    persistent P2P op lists) with all ranks as threads of this process; result == the single lattice."""
    import threading

    import phl
    from phl import rowtile
    from test_rowtile_cpu import make_image

    H, W, L = 64 * world, 48, 16
    feat, src = make_image(H, W, L, sigma_xy=3.0)
    dev = torch.device("cuda")
    fake = _LoopbackDist(world)
    outs, errs = {}, []

    def run(rank):
        try:
            fake.local.rank = rank
            job = rowtile.RowTileFilter(feat, L, rank, world, dev, fake, groups=groups)
            assert job._fused and len(job._ops) == groups
            assert job._edge_first == (groups == 1), job.describe()      # one group: edge chunks first, exchange under the rest
            mine = torch.from_numpy(src[job.row0 * W:(job.row0 + job.own_rows) * W]).to(dev)
            buf = torch.empty_like(mine)
            a = job.filter(mine, out=buf).clone()
            b = job.filter(mine)                       # second call: same buffers, same op lists
            assert torch.equal(a, b)
            outs[rank] = a.cpu().numpy()
        except Exception as e:      # noqa: BLE001
            import traceback

            errs.append((rank, traceback.format_exc()))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs, errs
    want = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev), reference_table=True).filter(torch.from_numpy(src).to(dev)).cpu().numpy()
    got = np.concatenate([outs[r] for r in range(world)], 0)
    assert rel(got, want) <= RTOL


@pytest.mark.parametrize("world", [8, 4])
def test_full_size_c3_in_row_bands(world):
    """BASELINE configs[3] at its real size: the 2048x1536x256 volume cut into 8 (and 4) row bands, all bands played
    on this one GPU (loopback exchange of exactly the rows RCCL would carry), against the single-lattice filter."""
    import os
    import sys

    import phl
    from phl import rowtile

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    H, W, L, _ = bench.WORKLOADS["c3"]
    feat = bench.synthetic_features(H, W)
    dev = torch.device("cuda")
    src = bench.synthetic_values(torch, H, W, L, 0, dev)
    # bands are cut out of the whole image's REFERENCE-TABLE lattice (rowtile.RowBand, table = "reference"): the result is
    # the reference's, duplicates of its hash table included -- not the defect-free lattice's
    want = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev), reference_table=True).filter(src)
    got, bands = rowtile.simulate(feat, src, world, phl.Lattice, dev)
    assert all(b.table == "reference" for b in bands)
    err = float(((got - want).abs() / want.abs().clamp_min(1e-3 * float(want.abs().max()))).max())
    rows = [b.recv_rows(p) for b in bands for p in b.sides]
    print(f"[measured] c3 in {world} bands: max rel err vs single lattice {err:.2e}; boundary rows per side "
          f"{min(rows)}..{max(rows)} = {max(rows) * L * 4 / 1e6:.1f} MB; ghosts per band "
          f"{max(b.M - b.eng.M_local for b in bands)} of {max(b.M for b in bands)} vertices")
    assert err <= RTOL
    assert all(b.own_rows == H // world for b in bands)


def test_splat_in_parts_equals_the_whole_splat():
    """phl_splat_part: the chunks that touch a set of vertex rows first (those rows complete), then the other chunks
    and rows -- bitwise the same vertex sums as one whole chunk splat, for any split."""
    import phl
    from test_rowtile_cpu import make_image

    feat, src = make_image(160, 128, 32, sigma_xy=3.0)
    dev = torch.device("cuda")
    L = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev))
    s = torch.from_numpy(src).to(dev)
    assert L.tile_stats(32)["staged_splat"] == 1
    whole = L.splat(s)
    rng = np.random.default_rng(0)
    keys = L.keys()
    rows_map = L.vertex_rows()
    for frac in (0.1, 0.5):
        pick = torch.from_numpy(np.sort(rng.permutation(L.M)[:int(L.M * frac)]).astype(np.int64)).to(dev)   # first-touch ids
        first_rows = rows_map[pick]
        mask = L.chunks_touching(first_rows)
        assert 0 < mask.sum() <= len(mask)
        is_first = torch.zeros(L.M, dtype=torch.bool, device=dev)
        is_first[first_rows] = True
        out = torch.full((L.M, 32), float("nan"), device=dev)
        partial = torch.empty((max(L.partial_rows, 1), 32), device=dev)
        ch = lambda m: torch.from_numpy(np.nonzero(m)[0].astype(np.int32)).to(dev)
        L.splat_part(s, out, partial, ch(mask), torch.nonzero(is_first).flatten().to(torch.int32))
        assert torch.equal(out[first_rows], whole[first_rows])          # complete before the second part runs
        L.splat_part(s, out, partial, ch(~mask), torch.nonzero(~is_first).flatten().to(torch.int32))
        assert torch.equal(out, whole)
    assert keys.shape[0] == L.M


def test_rccl_shaped_path_random_shapes():
    """Randomised images / rank counts / label counts through the device-payload driver (threads as ranks): the
    edge-first schedule where it applies (one group, chunk splat available), channel groups or the plain path
    elsewhere (L % 4 != 0, tiny bands) -- always equal to the single lattice."""
    import threading

    import phl
    from phl import rowtile
    from test_rowtile_cpu import make_image

    rng = np.random.default_rng(20261004)
    dev = torch.device("cuda")
    seen = set()
    for trial in range(6):
        world = int(rng.integers(2, 6))
        rows = int(rng.choice([48, 64, 96]))
        H, W = rows * world, int(rng.choice([40, 64, 100]))
        L = int(rng.choice([4, 6, 20, 64]))
        groups = int(rng.choice([1, 1, 2])) if L % 2 == 0 and L >= 4 else 1
        feat, src = make_image(H, W, L, sigma_xy=float(rng.choice([2.0, 3.0])))
        fake = _LoopbackDist(world)
        outs, errs = {}, []

        def run(rank):
            try:
                fake.local.rank = rank
                job = rowtile.RowTileFilter(feat, L, rank, world, dev, fake, groups=groups)
                mine = torch.from_numpy(src[job.row0 * W:(job.row0 + job.own_rows) * W]).to(dev)
                outs[rank] = (job.filter(mine).cpu().numpy(), job._edge_first)
            except Exception:      # noqa: BLE001
                import traceback

                errs.append((rank, traceback.format_exc()))

        ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=300)
        assert not errs, (trial, world, H, W, L, groups, errs)
        want = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev), reference_table=True).filter(torch.from_numpy(src).to(dev)).cpu().numpy()
        got = np.concatenate([outs[r][0] for r in range(world)], 0)
        assert rel(got, want) <= RTOL, (trial, world, H, W, L, groups)
        seen.add(any(outs[r][1] for r in range(world)))
    assert seen == {True, False}, seen          # both schedules were exercised


@pytest.mark.parametrize("workload", ["band8", "c5"])
def test_bench_two_ranks_validates_itself(workload):
    """`python bench.py --gpus 2`: the launcher starts two rank processes; on this one-GPU box they share the card
    and talk over gloo (the driver's 8-GPU node runs the same code over RCCL).  The JSON line must carry the run's
    own correctness check -- row bands against a single-lattice filter of the whole volume, independent volumes
    against the gather kernels -- and, for row bands, the exchange timing keys."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "3",
                        "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    chk = line["check"]
    assert chk["ok"] and chk["check_max_rel"] <= 1e-4
    if workload == "band8":
        assert line["scaling"] == "strong" and line["config"]["table"] == "reference"
        assert chk["vs_reference_table"]["rows_beyond_1e-4"] == 0.0
        for k in ("exchange_ms", "step_ms", "step_no_exchange_ms", "overlap_hidden_frac"):
            assert k in chk
    else:
        assert line["scaling"] == "weak" and chk["repeatable"]


def _run_ranks(feat, src, L, world, dev, **kw):
    """All ranks as threads over _LoopbackDist; returns ({rank: output}, {rank: job})."""
    import threading

    from phl import rowtile

    W = feat.shape[1]
    fake = _LoopbackDist(world)
    outs, jobs, errs = {}, {}, []

    def run(rank):
        try:
            fake.local.rank = rank
            job = rowtile.RowTileFilter(feat, L, rank, world, dev, fake, **kw)
            mine = src[job.row0 * W:(job.row0 + job.own_rows) * W]
            outs[rank] = job.filter(mine).clone()
            jobs[rank] = job
        except Exception:      # noqa: BLE001
            import traceback

            errs.append((rank, traceback.format_exc()))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not errs, errs
    return outs, jobs


@pytest.mark.parametrize("world,H,W,L,sigma", [(2, 256, 96, 64, 3.0), (3, 288, 200, 32, 2.0), (8, 1536, 256, 256, 8.0)])
def test_band_schedules_agree_bit_for_bit(world, H, W, L, sigma, monkeypatch):
    """The default band step -- edge chunks first, boundary rows summed AND packed by one kernel on a side stream, ghost
    rows received in place, every blur axis restricted to the rows something later reads -- against the same band
    computed the plain way (whole splat, gather, all rows blurred on every axis): the same bits on every own pixel.
    Restricting the blur is only legal because the rows it skips are never read; this is the test of that claim."""
    import bench

    feat = bench.synthetic_features(H, W, sigma_xy=sigma)
    dev = torch.device("cuda")
    src = bench.synthetic_values(torch, H, W, L, 0, dev)
    new, jobs = _run_ranks(feat, src, L, world, dev)
    inner = jobs[min(1, world - 1)]
    assert inner._edge_first and inner.band.blur_rows is not None
    rows = inner.describe()["rowtile"]["blur_rows_per_axis"]
    assert rows[-1] == inner.band.M_own and rows[0] >= rows[-1] and rows[0] <= inner.band.M       # the last axis: own rows only
    assert rows[0] > rows[-1], rows                                                                  # earlier axes carry ghosts
    monkeypatch.setenv("PHL_ROWTILE_BLUR_ROWS", "0")
    monkeypatch.setenv("PHL_ROWTILE_EDGE_FIRST", "0")
    old, jobs_old = _run_ranks(feat, src, L, world, dev, groups=1)
    assert not jobs_old[0]._edge_first and jobs_old[0].band.blur_rows is None
    for r in range(world):
        assert torch.equal(new[r], old[r]), f"rank {r}: schedules differ"
    monkeypatch.delenv("PHL_ROWTILE_EDGE_FIRST")
    mid, _ = _run_ranks(feat, src, L, world, dev)          # edge-first, unrestricted blur
    for r in range(world):
        assert torch.equal(new[r], mid[r]), f"rank {r}: restricted blur differs"
    monkeypatch.delenv("PHL_ROWTILE_BLUR_ROWS")
    for mode in ("edge, two queues", "whole"):             # the other two forms RowTileFilter.autotune chooses between
        monkeypatch.setenv("PHL_ROWTILE_MODE", mode)
        alt, jobs_alt = _run_ranks(feat, src, L, world, dev)
        assert jobs_alt[0]._mode == mode
        for r in range(world):
            assert torch.equal(new[r], alt[r]), f"rank {r}: mode '{mode}' differs"


def test_reduce_and_pack_in_one_kernel():
    """phl_splat_part_pack: the listed rows are completed and written to their places in a send buffer by one launch
    (sole-contributor rows are copied, multi-chunk rows summed, long lists reduced by a workgroup) == phl_splat_part
    followed by a gather."""
    import bench
    import phl

    feat = bench.tsukuba_features(288, 384, 0.1, 0.1)       # flat regions: vertices fed by long chunk lists
    dev = torch.device("cuda")
    L = 64
    lat = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev))
    assert lat.tile_stats(L)["staged_splat"] == 1
    src = torch.rand((feat.shape[0] * feat.shape[1], L), device=dev)
    rng = np.random.default_rng(3)
    pick = np.sort(rng.permutation(lat.M)[:lat.M // 3]).astype(np.int64)
    rows = torch.from_numpy(pick).to(dev)
    mask = lat.chunks_touching(rows)
    ch = torch.from_numpy(np.nonzero(mask)[0].astype(np.int32)).to(dev)
    partial = torch.empty((max(lat.partial_rows, 1), L), device=dev)
    a = torch.zeros((lat.M, L), device=dev)
    lat.splat_part(src, a, partial, ch, rows.to(torch.int32))
    want = lat.gather_rows(a, rows)
    perm = torch.from_numpy(rng.permutation(len(pick)).astype(np.int32)).to(dev)
    pack = torch.full((len(pick) + 3, L), float("nan"), device=dev)
    b = torch.zeros((lat.M, L), device=dev)
    none = torch.empty(0, dtype=torch.int32, device=dev)
    lat.splat_part(src, b, partial, ch, none)                                  # chunk sums only
    lat.splat_part(src, b, partial, none, rows.to(torch.int32), pack_pos=perm, pack=pack)     # rows + pack
    assert torch.equal(a[rows], b[rows])
    assert torch.equal(pack[perm.long()], want) and torch.isnan(pack[len(pick):]).all()


def test_band_cut_from_the_whole_lattice_is_its_restriction():
    """phl_sub_lattice: a band's lattice taken out of the whole image's reference-table lattice -- keys, the vertex every
    (pixel, remainder) resolves to, weights, and the blur neighbours (wherever the neighbour was selected too) are the whole
    lattice's, duplicates of the reference's hash table and their visibility included."""
    import bench
    import phl
    from phl import rowtile

    H, W = 256, 384
    feat = bench.synthetic_features(H, W, sigma_xy=3.0)
    dev = torch.device("cuda")
    whole = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev), reference_table=True)
    clean = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev))
    assert whole.M > clean.M > 16383, "this image should grow the reference's table and leave duplicate vertices"
    wk, (wvid, ww), wnb = whole.keys(), whole.replay(), whole.neighbors()
    r0, r1 = 96, 160
    p0, p1 = r0 * W, r1 * W
    own_mask = whole.vertices_of_pixels(p0, p1)
    assert np.array_equal(np.nonzero(own_mask)[0], np.unique(wvid[p0:p1]))
    own = np.nonzero(own_mask)[0]
    y = rowtile.vertex_coordinate(wk, 5, 1)
    lo, hi = feat[r0, 0, 1], feat[r1 - 1, 0, 1]
    ghosts = np.nonzero(~own_mask & (y >= lo - 3.01) & (y <= hi + 3.01))[0]
    ghosts = ghosts[np.argsort(np.maximum(lo - y[ghosts], y[ghosts] - hi), kind="stable")]
    sel = np.concatenate([own, ghosts]).astype(np.int32)
    sub = whole.sub_lattice(p0, p1, sel, len(own), torch.from_numpy(feat[r0:r1].reshape(-1, 5)).to(dev))
    assert sub.M == len(sel) and sub.M_local == len(own) and sub.n == p1 - p0
    assert np.array_equal(sub.keys(), wk[sel])
    svid, sw = sub.replay()
    assert np.array_equal(sel[svid], wvid[p0:p1]) and np.array_equal(sw, ww[p0:p1])
    rows = sub.vertex_rows().cpu().numpy()
    assert np.array_equal(rows[len(own):], np.arange(len(own), len(sel))), "ghost rows keep the caller's order behind the own rows"
    pos = np.full(whole.M, -1, np.int64)
    pos[sel] = np.arange(len(sel))
    snb = sub.neighbors()
    want = np.where(wnb[:, sel, :] >= 0, pos[np.clip(wnb[:, sel, :], 0, None)], -1)     # absent, or not selected: -1
    assert np.array_equal(snb, want)
    # ... and it filters: the band's own pixels, with the ghost rows fed from the whole lattice's splat, give the whole
    # lattice's output on those pixels
    L = 32
    src = torch.rand((H * W, L), device=dev)
    vw = whole.to_first_touch(whole.splat(src))[torch.from_numpy(sel.astype(np.int64)).to(dev)]       # selected vertices, first-touch order
    out = sub.slice(sub.blur(sub.from_first_touch(vw.contiguous())))
    ref = whole.filter(src)[p0:p1]
    # interior rows of the band (more than the blur's reach from vertices that were not selected) agree
    err = ((out - ref).abs() / ref.abs().clamp_min(1e-3 * float(ref.abs().max()))).max(dim=1).values.reshape(r1 - r0, W)
    assert float(err.max()) <= 1e-5, float(err.max())


def test_reference_table_bands_give_the_reference_result(monkeypatch):
    """Row bands cut from the whole reference-table lattice return the single reference-table lattice's result; bands
    built the old way (one defect-free lattice per band) differ from it on the rows the reference's duplicates touch."""
    import bench
    import phl

    H, W, L, world = 512, 384, 16, 4
    feat = bench.synthetic_features(H, W, sigma_xy=3.0)
    dev = torch.device("cuda")
    src = bench.synthetic_values(torch, H, W, L, 0, dev)
    whole = phl.Lattice(torch.from_numpy(feat.reshape(-1, 5)).to(dev), reference_table=True)
    want = whole.filter(src)

    def far_rows(got):
        rel_ = ((got - want).abs() / want.abs().clamp_min(1e-3 * float(want.abs().max()))).max(dim=1).values
        return float((rel_ > 1e-4).float().mean()), float(rel_.max())

    outs, jobs = _run_ranks(feat, src, L, world, dev)
    assert all(j.band.table == "reference" for j in jobs.values())
    got = torch.cat([outs[r] for r in range(world)], 0)
    frac, mx = far_rows(got)
    print(f"[measured] reference-table bands vs the single reference-table lattice: max rel {mx:.2e}")
    assert frac == 0.0 and mx <= 1e-5
    monkeypatch.setenv("PHL_ROWTILE_TABLE", "clean")
    outs_c, jobs_c = _run_ranks(feat, src, L, world, dev)
    assert all(j.band.table == "clean" for j in jobs_c.values())
    frac_c, mx_c = far_rows(torch.cat([outs_c[r] for r in range(world)], 0))
    print(f"[measured] defect-free bands vs the reference-table lattice: {frac_c:.5f} of the rows beyond 1e-4 (max rel {mx_c:.2e})")
    assert frac_c > 0.0


def test_sub_lattice_edge_cases():
    """phl_sub_lattice argument handling: a selection that misses a vertex the band's pixels touch is refused (never a
    silent wrong lattice), a selection without ghosts is a lattice of the band alone, ghost-only tails keep their order."""
    import phl
    from test_rowtile_cpu import make_image

    feat, src = make_image(64, 48, 8, sigma_xy=3.0)
    dev = torch.device("cuda")
    W = 48
    whole = phl.Lattice.whole_image(torch.from_numpy(feat.reshape(-1, 5)).to(dev))
    p0, p1 = 16 * W, 40 * W
    own = np.nonzero(whole.vertices_of_pixels(p0, p1))[0].astype(np.int32)
    band_feat = torch.from_numpy(feat[16:40].reshape(-1, 5)).to(dev)
    with pytest.raises(phl.PhlError) as ei:
        whole.sub_lattice(p0, p1, own[:-1], len(own) - 1, band_feat)          # one own vertex short
    assert ei.value.status == 1
    with pytest.raises(phl.PhlError):
        whole.sub_lattice(p0, p1, own, len(own) + 1, band_feat)               # n_own > n_sel
    with pytest.raises(phl.PhlError):
        whole.sub_lattice(p0, p1, np.concatenate([own, own[:1]]), len(own), band_feat)    # a repeated vertex
    alone = whole.sub_lattice(p0, p1, own, len(own), band_feat)
    assert alone.M == alone.M_local == len(own)
    s = torch.from_numpy(src[p0:p1]).to(dev)
    # a band without ghosts filters like a lattice built from the band's own pixels (same vertices: M below the first doubling)
    ref = phl.Lattice(band_feat).filter(s)
    got = alone.filter(s)
    assert whole.M < 16383 and float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    rest = np.setdiff1d(np.arange(whole.M, dtype=np.int32), own)[::-1].copy()   # every other vertex as a ghost, descending ids
    sub = whole.sub_lattice(p0, p1, np.concatenate([own, rest]), len(own), band_feat)
    assert sub.M == whole.M and np.array_equal(sub.keys()[len(own):], whole.keys()[rest])
    rows = sub.vertex_rows().cpu().numpy()
    assert np.array_equal(rows[len(own):], np.arange(len(own), whole.M))


def test_mean_field_inference_on_row_bands():
    """mean_field_infer (crf/crf_module.py:41-53) with the row-band operator as its W: every rank iterates on its own rows of
    E_0 (compatibility product and softmax are per pixel, only W @ Q exchanges boundary vertices); the ranks' results stacked
    are the single-GPU result of the mirrored API on the whole image."""
    import threading

    import bench
    import crf.crf_module as cm
    from crf.gaussian_matrix import LatticeGaussian
    from phl import rowtile

    H, W, L, world, niters = 384, 128, 32, 3, 3
    feat = bench.synthetic_features(H, W, sigma_xy=4.0)
    dev = torch.device("cuda")
    E0 = torch.rand((H * W, L), device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * 10.0
    labels = torch.arange(L, dtype=torch.float32, device=dev)
    Mu = cm.compatibility_matrix(lambda a, b: cm.charbonneir(a, b, 3.0), labels)
    want = cm.mean_field_infer(E0, LatticeGaussian(torch.from_numpy(feat.reshape(-1, 5)).to(dev)), Mu, niters)
    fake = _LoopbackDist(world)
    outs, errs = {}, []

    def run(rank):
        try:
            fake.local.rank = rank
            Wb = rowtile.RowBandGaussian(feat, L, rank, world, dev, fake)
            outs[rank] = cm.mean_field_infer(Wb.rows(E0).contiguous(), Wb, Mu, niters).clone()
        except Exception:      # noqa: BLE001
            import traceback

            errs.append((rank, traceback.format_exc()))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not errs, errs
    got = torch.cat([outs[r] for r in range(world)], 0)
    err = float(((got - want).abs() / want.abs().clamp_min(1e-3 * float(want.abs().max()))).max())
    disp = float(((got @ labels - want @ labels).abs() / (want @ labels).abs().clamp_min(1e-2)).max())
    print(f"[measured] mean field on {world} row bands vs one GPU, {niters} iterations: Q rel {err:.2e}, disparity rel {disp:.2e}")
    assert err <= 1e-4 and disp <= 1e-5


class _AsyncLoopbackDist(_LoopbackDist):
    """_LoopbackDist with RCCL's stream semantics instead of a blocking exchange: a batch is ENQUEUED on the rank's
    communication stream behind the caller's current stream (as ProcessGroupNCCL orders its work), runs late (a spin kernel in
    front of every batch widens every race window), and Req.wait() only makes the caller's current stream wait for it -- the
    host never blocks on the GPU.  A schedule that reads a receive buffer, or overwrites a send buffer, without the stream
    dependency RCCL needs shows up as wrong numbers here.
    Messages travel through PERSISTENT staging buffers, one per (sender, receiver, position in the batch), handed back and
    forth with events: nothing is allocated after the first step (an allocation may synchronise the device, which would
    hide exactly the races this harness is for, and a block freed on one stream while another still reads it is a race of
    the harness's own)."""

    class _Req:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)
            return True

    def __init__(self, world, delay_cycles=2_000_000):
        import queue

        super().__init__(world)
        self.comm = {}
        self.delay = delay_cycles
        self.stage = {}                               # (src, dst, k) -> staging tensor
        self.read_done = {(a, b): queue.Queue() for a in range(world) for b in range(world)}   # receiver -> sender: (k, event)

    def batch_isend_irecv(self, ops):
        me = self.local.rank
        comm = self.comm.get(me)
        if comm is None:
            comm = self.comm[me] = torch.cuda.Stream()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        comm.wait_event(ev)
        with torch.cuda.stream(comm):
            torch.cuda._sleep(self.delay)                        # the wire is slow
            k_of = {}
            for o in ops:
                if o.op == "isend":
                    k = k_of[o.peer] = k_of.get(o.peer, -1) + 1
                    key = (me, o.peer, k)
                    buf = self.stage.get(key)
                    if buf is None:
                        buf = self.stage[key] = torch.empty_like(o.tensor)
                    else:                                       # its previous content has been read (the receiver says when)
                        kk, rd = self.read_done[(me, o.peer)].get(timeout=300)
                        assert kk == k and buf.shape == o.tensor.shape
                        comm.wait_event(rd)
                    buf.copy_(o.tensor)
                    e = torch.cuda.Event()
                    e.record(comm)
                    self.q[(me, o.peer)].put((k, buf, e))
            for o in ops:
                if o.op == "irecv":
                    k, t, e = self.q[(o.peer, me)].get(timeout=300)
                    comm.wait_event(e)
                    if o.tensor.dtype == torch.uint8 or t.dtype == torch.uint8:
                        o.tensor.view(torch.uint8).copy_(t.view(torch.uint8))
                    else:
                        o.tensor.copy_(t)
                    rd = torch.cuda.Event()
                    rd.record(comm)
                    self.read_done[(o.peer, me)].put((k, rd))
            done = torch.cuda.Event()
            done.record(comm)
        return [self._Req(done)]


@pytest.mark.parametrize("groups", [None, 2])
def test_edge_first_schedule_under_asynchronous_exchange(groups):
    """The band step (groups=None: edge-first, ghost rows received in place, send buffer packed by the reduction; groups=2: the
    channel groups pipelined) with an exchange
    that behaves like RCCL -- enqueued, late, never blocking the host -- over several back-to-back steps with changing inputs:
    every step's result equals the one computed with the blocking loopback exchange, bit for bit."""
    import threading

    import bench
    from phl import rowtile

    H, W, L, world, steps = 768, 256, 64, 4, 4
    feat = bench.synthetic_features(H, W, sigma_xy=6.0)
    dev = torch.device("cuda")
    srcs = [bench.synthetic_values(torch, H, W, L, 100 * s, dev) for s in range(steps)]

    def run_all(fake):
        outs, errs = {}, []

        def run(rank):
            try:
                fake.local.rank = rank
                job = rowtile.RowTileFilter(feat, L, rank, world, dev, fake, groups=groups)
                assert job._direct and (job._edge_first if groups is None else (job._fused and not job._edge_first))
                res = []
                for s in range(steps):          # no host synchronisation between steps: buffers are reused while work is in flight
                    res.append(job.filter(srcs[s][job.row0 * W:(job.row0 + job.own_rows) * W]))
                torch.cuda.synchronize()
                outs[rank] = res
            except Exception:      # noqa: BLE001
                import traceback

                errs.append((rank, traceback.format_exc()))

        ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=600)
        assert not errs, errs
        return outs

    want = run_all(_LoopbackDist(world))
    got = run_all(_AsyncLoopbackDist(world))
    for r in range(world):
        for s in range(steps):
            assert torch.equal(got[r][s], want[r][s]), f"rank {r} step {s}: the asynchronous exchange changes the result"
    # The harness is what it claims to be -- checked directly rather than through the outcome of a deliberately broken run
    # (whether a missing wait shows in the numbers depends on which hardware queue the streams happen to share: with the
    # communication stream's spin kernel ahead of the caller's kernels in one queue the broken run is accidentally ordered):
    # a batch returns to the host while its exchange is still pending, and only wait() + the stream deliver the data.
    probe = _AsyncLoopbackDist(2, delay_cycles=60_000_000)      # ~30 ms on the wire
    seen, perr = {}, []

    def ping(rank):
        try:
            probe.local.rank = rank
            mine = torch.full((1024,), float(rank + 1), device=dev)
            theirs = torch.zeros((1024,), device=dev)
            reqs = probe.batch_isend_irecv([probe.P2POp(probe.isend, mine, 1 - rank), probe.P2POp(probe.irecv, theirs, 1 - rank)])
            pending = not reqs[0].ev.query()                    # the host is back, the exchange is not done
            for r in reqs:
                r.wait()
            torch.cuda.current_stream().synchronize()
            seen[rank] = (pending, float(theirs[0]), float(theirs[-1]))
        except Exception:      # noqa: BLE001
            import traceback

            perr.append(traceback.format_exc())

    ts = [threading.Thread(target=ping, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not perr, perr
    assert seen == {0: (True, 2.0, 2.0), 1: (True, 1.0, 1.0)}, seen
