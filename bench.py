#!/usr/bin/env python3
"""bench.py -- throughput of the permutohedral mean-field message-passing step on MI355X.

One "step" = one pass of the hot path (splat -> blur over d+1 axes -> slice) over the whole
synthetic H x W x L value volume, inputs already resident in HBM, lattice built once before
the timed region (init-once / filter-many; the build time is reported next to it).

Metric (BASELINE.json): Mpixel-labels/s per CRF mean-field iter = H*W*L / t(step) / 1e6.
Default workload = the volume the north-star target is quoted on: 2048 x 1536 x 256, d = 5
bilateral features (SURVEY.md section 8d synthetic recipe).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c5|c1]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1 shards the image by row bands over the ranks (strong scaling, same total volume); see
depth-estimation_amd/phl/rowtile.py for the halo exchange over RCCL.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd"))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (H, W, L, description)
    "c3": (1536, 2048, 256, "synthetic 2048x1536x256 cost volume, d=5 bilateral features (BASELINE configs[2])"),
    "c2": (1110, 1390, 256, "Middlebury-sized 1390x1110x256, d=5 (BASELINE configs[1], synthetic features)"),
    "c5": (1024, 1024, 128, "1024x1024x128 volume, d=5 (BASELINE configs[4], one volume per GPU)"),
    "c1": (288, 384, 16, "Tsukuba-sized 384x288x16, d=5 (BASELINE configs[0])"),
    "band8": (192, 2048, 256, "one eighth of c3 (192 rows): the per-rank share of an 8-GPU row-band run, for overhead studies"),
}
SIGMA_XY, SIGMA_C = 8.0, 0.1
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def ensure_built(local_rank):
    """The HIP library normally travels with the tree (built by __graft_entry__.build()).  If it did not,
    local rank 0 compiles it here (hipcc, ~1 min) and the other ranks wait; never a CPU fallback."""
    lib = os.path.join(ROOT, "depth-estimation_amd", "lib", "libphl.so")
    if os.path.exists(lib):
        return
    if local_rank == 0:
        import subprocess

        print("bench.py: libphl.so missing, building it with hipcc", file=sys.stderr)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "depth-estimation_amd", "csrc"), "-j4", "ARCH=gfx950"],
                              stdout=sys.stderr)
    else:
        t0 = time.time()
        while not os.path.exists(lib):
            if time.time() - t0 > 900:
                sys.exit("bench.py: libphl.so did not appear (local rank 0 builds it)")
            time.sleep(1.0)
        time.sleep(2.0)     # let the linker finish writing


def box_blur(a, r):
    """(2r+1)^2 box mean with edge replication, via cumulative sums."""
    for axis in (0, 1):
        pad = [(0, 0)] * a.ndim
        pad[axis] = (r + 1, r)
        c = np.cumsum(np.pad(a, pad, mode="edge"), axis=axis, dtype=np.float64)
        hi = [slice(None)] * a.ndim
        lo = [slice(None)] * a.ndim
        hi[axis] = slice(2 * r + 1, None)
        lo[axis] = slice(0, -(2 * r + 1))
        a = ((c[tuple(hi)] - c[tuple(lo)]) / (2 * r + 1)).astype(np.float32)
    return a


def synthetic_features(H, W, sigma_xy=SIGMA_XY, sigma_c=SIGMA_C):
    """x=col/sigma_xy, y=row/sigma_xy (pixels), 3 channels of N(0,1) noise
    box-blurred r=16 twice, rescaled to [0,1], / sigma_c.  Returns [H, W, 5] fp32."""
    rng = np.random.default_rng(1234)
    col = rng.standard_normal((H, W, 3)).astype(np.float32)
    col = box_blur(box_blur(col, 16), 16)
    col -= col.min(axis=(0, 1), keepdims=True)
    col /= col.max(axis=(0, 1), keepdims=True)
    feat = np.empty((H, W, 5), np.float32)
    feat[..., 0] = (np.arange(W, dtype=np.float32) / sigma_xy)[None, :]
    feat[..., 1] = (np.arange(H, dtype=np.float32) / sigma_xy)[:, None]
    feat[..., 2:] = col / sigma_c
    return feat


def synthetic_values(torch, rows, W, L, row0, device):
    """Q = softmax(-U[0,10)) over L, pixel-major [rows*W, L]; seeded per image row so that any
    row band of the image is reproducible on any rank."""
    g = torch.Generator(device=device)
    out = torch.empty((rows * W, L), dtype=torch.float32, device=device)
    chunk = 64
    for r in range(0, rows, chunk):
        k = min(chunk, rows - r)
        g.manual_seed(4321 + row0 + r)
        u = torch.rand((k * W, L), generator=g, device=device) * 10.0
        out[r * W:(r + k) * W] = torch.softmax(-u, dim=1)
    return out


def algorithmic_bytes(n, M, L, d):
    """SURVEY.md 8(d), fp32, per launch."""
    return {
        "splat": 4 * n * L + 8 * n * (d + 1) + 4 * M * L,
        "blur": (d + 1) * (4 * M * L + 4 * M * L + 8 * M),     # d+1 axes; run as ceil((d+1)/2) fused launches
        "slice": 4 * M * L + 8 * n * (d + 1) + 4 * n * L,
    }


def pmc_traffic(kernel_names, workload):
    """HBM bytes per launch of the named kernel(s) from the committed rocprofv3 --pmc summary
    (counters cannot be collected from inside the timed process; FETCH_SIZE is doubled there as
    MI355X_MICROARCH.md prescribes for gfx950).  Only valid for the workload it was taken on."""
    path = os.path.join(ROOT, "profiles", "r01_final_pmc_traffic.json")
    if workload != "c3" or not os.path.exists(path):
        return None, None
    k = json.load(open(path))["kernels"]
    total = 0
    for name in kernel_names.split("+"):
        count, _, name = name.rpartition("*")
        hit = [v for kk, v in k.items() if kk.split("<")[0] == name]
        if not hit:
            return None, None
        total += int(count or 1) * hit[0]["hbm_bytes_per_launch"]
    return int(total), "profiles/r01_final_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, same command)"


def cpu_baseline(feat, H, W, L, d, torch):
    """Reference engine (oracle/_ref, the reference's own C++) or our C port, one thread, on a
    bounded crop of the same workload.  Reported beside the GPU number; not a target."""
    from oracle import phl_oracle as po

    ch, cw = min(H, 1024), min(W, 1536)      # ~10 s of single-thread CPU work at L=256
    ref = np.ascontiguousarray(feat[:ch, :cw].reshape(-1, d))
    rng = np.random.default_rng(4321)
    src = rng.random((ch * cw, L), dtype=np.float32)
    src /= src.sum(1, keepdims=True)
    sample = f"top-left {cw}x{ch} crop of the same features, L={L}, 1 thread, lattice rebuilt per call (reference behaviour)"
    if po.reference_available():
        t0 = time.time()
        _, M, st = po.reference_filter_timed(src, ref)
        dt = time.time() - t0
        kind, stages = "reference", dict(init=st[0], splat=st[1], blur=st[2], slice=st[3])
    else:
        po.build_oracle()
        t0 = time.time()
        O = po.Oracle(ref)
        tb = time.time() - t0
        _, st = O.filter(src, timing=True)
        dt = time.time() - t0
        M = O.M
        kind, stages = "port", dict(build=tb, splat=st[0], blur=st[1], slice=st[2])
    return {"value": round(ch * cw * L / dt / 1e6, 3), "unit": "Mpixel-labels/s", "cores": 1, "kind": kind,
            "sample": sample, "seconds": round(dt, 3), "M_over_n": round(M / (ch * cw), 4),
            "stage_seconds": {k: round(float(v), 4) for k, v in stages.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tiles", action="store_true", help="plain gather kernels (A/B against the LDS-staged chunk path)")
    ap.add_argument("--exact", action="store_true", help="reference-exact arithmetic (bit-identical to the CPU path)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph (launch-bound sizes)")
    ap.add_argument("--force-rowtile", action="store_true", help="debug: run the row-band driver even with one rank")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py: no HIP device; the lattice filter has no CPU path to benchmark")
    ensure_built(local_rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_rowtile:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL peer-to-peer needs on this pool
        if not (os.environ.get("MASTER_ADDR") and os.environ.get("MASTER_PORT")):
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import phl

    H, W, L, desc = WORKLOADS[args.workload]
    d = 5
    feat = synthetic_features(H, W)
    n_total = H * W

    rowtiled = world > 1 or args.force_rowtile
    if rowtiled and args.workload == "c5":
        rowtiled = False      # BASELINE configs[4]: independent volumes, one per GPU, no exchange
    if rowtiled:
        from phl import rowtile

        job = rowtile.RowTileFilter(feat, L, rank, world, device, dist)
        src = synthetic_values(torch, job.own_rows, W, L, job.row0, device)
        step = lambda: job.filter(src)
        build_ms, M, n_local = job.build_ms, job.M, job.n_local
        extra = job.describe()
    else:
        ref = torch.from_numpy(feat.reshape(-1, d)).to(device)
        src = synthetic_values(torch, H, W, L, 0, device)
        torch.cuda.synchronize()
        t0 = time.time()
        lat = phl.Lattice(ref)
        torch.cuda.synchronize()
        build_ms = (time.time() - t0) * 1e3
        t0 = time.time()
        lat_warm = phl.Lattice(ref)          # second build: work arrays come from the cached scratch block
        torch.cuda.synchronize()
        build_warm_ms = (time.time() - t0) * 1e3
        del lat_warm
        lat.reserve(L)
        out = torch.empty_like(src)
        kw = dict(exact=args.exact, no_tiles=args.no_tiles)
        step = lambda: lat.filter(src, out=out, **kw)
        M, n_local = lat.M, n_total
        extra = {"tiles": lat.tile_stats(L), "lattice_build_warm_ms": round(build_warm_ms, 2)}

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    if args.graph and not rowtiled:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        step = graph.replay
        step()
        sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    volumes = world if (world > 1 and not rowtiled) else 1     # independent volumes: one per rank
    value = volumes * n_total * L / (dt / args.steps) / 1e6

    # ---- per-kernel timing with HIP events on the launch stream (rank-local lattice) ---------
    roofline = None
    stage_ms = {}
    if rowtiled:
        # rank 0's own band (own pixels + ghost vertices): the same kernels, timed locally
        lat = job.band.eng
        out = torch.empty_like(src)
        kw = dict(exact=False, no_tiles=False)
        extra["tiles"] = lat.tile_stats(L)
    if rank == 0 or not rowtiled:
        ev = lambda: torch.cuda.Event(enable_timing=True)
        reps = max(3, min(args.steps, 10))
        acc = {"splat": 0.0, "blur": 0.0, "slice": 0.0}
        scratch = None
        for _ in range(reps):
            e = [ev() for _ in range(4)]
            e[0].record()
            v = lat.splat(src, **kw)
            e[1].record()
            scratch = torch.empty_like(v) if scratch is None else scratch
            a = lat.blur(v, scratch)                 # all d+1 axes, two per launch
            e[2].record()
            lat.slice(a, out=out, **kw)
            e[3].record()
            torch.cuda.synchronize()
            for k, name in enumerate(("splat", "blur", "slice")):
                acc[name] += e[k].elapsed_time(e[k + 1])
            del v, a
        stage_ms = {k: v / reps for k, v in acc.items()}
        blur_launches = (d + 2) // 2
        dom = max(stage_ms, key=stage_ms.get)
        ab = algorithmic_bytes(n_local, M, L, d)
        achieved = ab[dom] / (stage_ms[dom] * 1e-3) / 1e9
        staged = extra["tiles"]["staged_splat"] and not (args.no_tiles or args.exact)
        staged_sl = extra["tiles"]["staged_slice"] and not args.no_tiles
        kname = {"splat": "k_splat_tiled+k_splat_reduce" if staged else "k_splat",
                 "blur": f"{(d + 1) // 2}*k_blur2" + ("+k_blur" if (d + 1) % 2 else ""),
                 "slice": "k_slice_tiled" if staged_sl else "k_slice"}
        traffic, traffic_src = pmc_traffic(kname[dom], args.workload) if not rowtiled else (None, None)
        # measured streaming ceiling on this box: device copy of the value volume, R+W bytes
        e0, e1 = ev(), ev()
        phl.stream_copy(out, src)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            phl.stream_copy(out, src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 2 * src.numel() * 4 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": kname[dom],
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": int(ab[dom]), "avg_launch_ms": round(stage_ms[dom], 4),
                    "per_stage": {k: {"ms": round(stage_ms[k], 4), "launches_per_step": (blur_launches if k == "blur" else 1),
                                      "algorithmic_GBps": round(ab[k] / (stage_ms[k] * 1e-3) / 1e9, 1)} for k in stage_ms},
                    # north-star wording: blur-pass READ bytes (d+1)*4*M*L against the HBM-read roofline
                    "measured_copy_GBps": round(copy_gbs, 1), "frac_of_measured_copy": round(achieved / copy_gbs, 4),
                    "blur_read_frac_of_peak": round((d + 1) * 4 * M * L / (stage_ms["blur"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if rowtiled:
            roofline["scope"] = f"rank 0's band only ({n_local} pixels, {M} vertices incl. ghosts); kernels as in the 1-GPU run"

    cpu = None
    if rank == 0 and world == 1 and not rowtiled and not args.no_cpu_baseline:
        cpu = cpu_baseline(feat, H, W, L, d, torch)

    if rank == 0:
        line = {
            "metric": "Mpixel-labels/s per CRF mean-field iter (splat+blur+slice)",
            "value": round(value, 1), "unit": "Mpixel-labels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if (world > 1 and rowtiled) else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "H": H, "W": W, "L": L, "d": d, "sigma_xy": SIGMA_XY,
                       "sigma_c": SIGMA_C, "n": n_total, "M": int(M), "M_over_n": round(M / n_local, 4),
                       "parallelism": ("single GPU" if world == 1 and not rowtiled else
                                       f"row bands x{world} + RCCL boundary-vertex exchange" if rowtiled else
                                       f"{world} independent volumes, one per GPU, no collective"),
                       "launch": "hip graph replay" if (args.graph and not rowtiled) else "eager",
                       "arithmetic": "reference-exact (bit-identical to the CPU path)" if args.exact else "default (fp32-rounding-equivalent, ~1e-7 rel)"},
            "lattice_build_ms": round(build_ms, 2),
            # the reference rebuilds its lattice in every filter call: the same metric with a (steady-state, warm
            # scratch) build added to every step -- SURVEY 8d asks for both
            "value_rebuild_each_iter": round(volumes * n_total * L / ((ms_per_step + extra.get("lattice_build_warm_ms", build_ms)) * 1e-3) / 1e6, 1),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        line.update(extra)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
