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
depth-estimation_amd/phl/rowtile.py for the halo exchange over RCCL.  `--workload c5` instead
gives every rank its own volume (BASELINE configs[4]; the reference's one-image-per-worker
batch mode, crf/gaussian_matrix.py:370-377), no data-path collective, weak scaling.

Launch forms (both supported):
  * `python bench.py --gpus N ...` with no torchrun environment: this process becomes a
    LAUNCHER -- it never touches the GPU (no torch import), starts N fresh rank processes of
    this same file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, forwards rank 0's JSON
    line and exits with the worst child's code;
  * `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`: RANK etc.
    come from the environment, this process is one rank.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# ROCr reads this at hsa_init, i.e. before the first torch.cuda / HIP call of the process: RCCL's
# peer-to-peer transport needs dmabuf IPC on this pool.  Must precede `import torch`.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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
JSON_OUT = sys.stdout
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


def synthetic_features(H, W, sigma_xy=SIGMA_XY, sigma_c=SIGMA_C, iid=False):
    """x=col/sigma_xy, y=row/sigma_xy (pixels), 3 channels of N(0,1) noise
    box-blurred r=16 twice, rescaled to [0,1], / sigma_c.  Returns [H, W, 5] fp32.
    iid=True: the 8(d) stress case, colours iid U[0,1] (every pixel in its own simplex, M/n -> d+1)."""
    rng = np.random.default_rng(1234)
    if iid:
        col = rng.random((H, W, 3), dtype=np.float32)
    else:
        col = rng.standard_normal((H, W, 3)).astype(np.float32)
        col = box_blur(box_blur(col, 16), 16)
        col -= col.min(axis=(0, 1), keepdims=True)
        col /= col.max(axis=(0, 1), keepdims=True)
    feat = np.empty((H, W, 5), np.float32)
    feat[..., 0] = (np.arange(W, dtype=np.float32) / sigma_xy)[None, :]
    feat[..., 1] = (np.arange(H, dtype=np.float32) / sigma_xy)[:, None]
    feat[..., 2:] = col / sigma_c
    return feat


def tsukuba_features(H, W, sigma_c, sigma_p):
    """Natural-image features in the reference notebook's own scaling (Experiments/DenseCrf.ipynb:142-146,
    crf/lattice/lite/test_bilateral.ipynb cell 6): (rgb / sigma_c, ij / sqrt(h^2 + w^2) / sigma_p), the image being
    the 384x288 Tsukuba frame stored as DATA in tests/golden/growth_tsukuba_384x288_vd4.npz (`img_u8`), bicubically
    upsampled to H x W (the Middlebury-sized stand-in; the datasets themselves are absent).
    Returns [H, W, 5] fp32 in the notebook's order (r, g, b, i, j)."""
    import torch
    import torch.nn.functional as F

    z = np.load(os.path.join(ROOT, "tests", "golden", "growth_tsukuba_384x288_vd4.npz"))
    img = torch.from_numpy(z["img_u8"].astype(np.float32) / 255.0).permute(2, 0, 1)[None]
    if tuple(img.shape[2:]) != (H, W):
        img = F.interpolate(img, size=(H, W), mode="bicubic", align_corners=False).clamp_(0, 1)
    feat = np.empty((H, W, 5), np.float32)
    feat[..., :3] = img[0].permute(1, 2, 0).numpy() / sigma_c
    diag = float(np.sqrt(H * H + W * W))
    feat[..., 3] = (np.arange(H, dtype=np.float32) / diag / sigma_p)[:, None]
    feat[..., 4] = (np.arange(W, dtype=np.float32) / diag / sigma_p)[None, :]
    return feat


def xyd_features(H, W, sigma_xy=SIGMA_XY, sigma_d=1.0, max_disp=64.0):
    """The north star's other feature set, (x, y, disparity): d = 3.  Disparity = a smooth field in [0, max_disp]
    (one channel of the SURVEY 8d noise recipe) / sigma_d.  Returns [H, W, 3] fp32."""
    rng = np.random.default_rng(4242)
    disp = box_blur(box_blur(rng.standard_normal((H, W, 1)).astype(np.float32), 16), 16)
    disp -= disp.min()
    disp *= max_disp / disp.max()
    feat = np.empty((H, W, 3), np.float32)
    feat[..., 0] = (np.arange(W, dtype=np.float32) / sigma_xy)[None, :]
    feat[..., 1] = (np.arange(H, dtype=np.float32) / sigma_xy)[:, None]
    feat[..., 2] = disp[..., 0] / sigma_d
    return feat


def features_for(H, W, sigma_xy=SIGMA_XY, sigma_c=SIGMA_C, iid=False, tsukuba=None, xyd=None):
    """One place that turns the bench's feature options into [H, W, d] features and a short label."""
    if xyd is not None:
        return xyd_features(H, W, sigma_xy, xyd), f"(x, y, disparity) d=3 sigma_xy={sigma_xy} sigma_d={xyd}"
    if tsukuba is not None:
        sc, sp = tsukuba
        return tsukuba_features(H, W, sc, sp), f"tsukuba-upsampled sigma_c={sc} sigma_p={sp}"
    return (synthetic_features(H, W, sigma_xy, sigma_c, iid),
            f"synthetic {'iid' if iid else 'smooth'} colours sigma_xy={sigma_xy} sigma_c={sigma_c}")


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


def stage_times(torch, lat, src, out, kw, reps):
    """HIP-event times (ms) of the three stages on the launch stream, averaged over `reps` passes."""
    ev = lambda: torch.cuda.Event(enable_timing=True)
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
    return {k: v / reps for k, v in acc.items()}


def algorithmic_bytes(n, M, L, d):
    """SURVEY.md 8(d), fp32, per launch."""
    return {
        "splat": 4 * n * L + 8 * n * (d + 1) + 4 * M * L,
        "blur": (d + 1) * (4 * M * L + 4 * M * L + 8 * M),     # d+1 axes; run as ceil((d+1)/2) fused launches
        "slice": 4 * M * L + 8 * n * (d + 1) + 4 * n * L,
    }


def hot_source_sha():
    """sha256 (first 16 hex digits) over the sources of the hot kernels: stored next to the PMC traffic figures when they
    are condensed (tools/save_profiles.py), so that a bench line can tell whether the committed counters were taken on the
    kernels it has just timed."""
    import hashlib

    h = hashlib.sha256()
    for name in ("phl_tiles.hip", "phl_filter.hip", "phl_device_utils.h"):
        with open(os.path.join(ROOT, "depth-estimation_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_names, workload):
    """HBM bytes per launch of the named kernel(s) from the committed rocprofv3 --pmc summary
    (counters cannot be collected from inside the timed process; FETCH_SIZE is doubled there as
    MI355X_MICROARCH.md prescribes for gfx950).  Only valid for the workload it was taken on."""
    import glob

    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))     # newest round last
    if workload != "c3" or not found:
        return None, None
    path = found[-1]
    doc = json.load(open(path))
    k = doc["kernels"]
    stored = doc.get("hot_kernel_source_sha")
    stale = None if stored is None else (stored != hot_source_sha())
    total = 0
    for name in kernel_names.split("+"):
        count, _, name = name.rpartition("*")
        hit = [v for kk, v in k.items() if kk.split("<")[0] == name]
        if not hit:
            return None, None
        total += int(count or 1) * hit[0]["hbm_bytes_per_launch"]
    note = ("; TAKEN ON OTHER KERNEL SOURCES than this run's (hot_kernel_source_sha differs): refresh with tools/profile_round.sh"
            if stale else ("" if stale is False else "; source hash of the profiled kernels not recorded"))
    # a figure measured on other kernels than the ones just timed is not reported as this run's traffic
    return (None if stale else int(total)), f"profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, same command){note}"


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_filter_once(src, ref, want_out=False):
    """One reference-shaped call (lattice rebuilt inside, as the reference does) on the CPU checker:
    the reference's own engine when oracle/_ref travelled with the tree, else our C port."""
    from oracle import phl_oracle as po

    if po.reference_available():
        t0 = time.time()
        out, M, st = po.reference_filter_timed(src, ref)
        res = ("reference", time.time() - t0, int(M), dict(init=st[0], splat=st[1], blur=st[2], slice=st[3]))
        return res + (out,) if want_out else res
    po.build_oracle()
    t0 = time.time()
    O = po.Oracle(ref, faithful_table=True)       # the reference's table behaviour across its doublings
    tb = time.time() - t0
    out, st = O.filter(src, timing=True)
    res = ("port", time.time() - t0, int(O.M), dict(build=tb, splat=st[0], blur=st[1], slice=st[2]))
    return res + (out,) if want_out else res


def cpu_worker(path):
    """Child of cpu_baseline()'s batch leg: a fresh process (numpy + the CPU checker only, no torch, no GPU)
    that filters one stored item -- the reference's `mp.Pool(num_threads).starmap(latticefilter, ...)`
    worker (crf/gaussian_matrix.py:370-377)."""
    z = np.load(path)
    kind, dt, M, _ = _cpu_filter_once(z["src"], z["ref"])[:4]
    print(json.dumps({"kind": kind, "seconds": dt, "M": M}))


def gpu_vs_reference(ref, src, want, M_ref):
    """Bar (1) at BASELINE scale, checked inside the driver's own bench run: the crop the CPU leg just filtered
    with the reference engine (M ~ 150 k: four doublings of the reference's hash table, permutohedral.h:59-62,
    101-103) is filtered on the GPU and compared with that output.
      * reference-table build + exact arithmetic: must be BIT-IDENTICAL;
      * reference-table build + default arithmetic: fp32 rounding only;
      * clean-table build (one vertex per key; the reference's duplicates are a defect of its table): the rows that
        differ beyond 1e-4 are the ones the duplicates touch."""
    import torch

    import phl

    dev = torch.device("cuda", torch.cuda.current_device())
    r = torch.from_numpy(ref).to(dev)
    s = torch.from_numpy(src).to(dev)
    res = {}
    scale = np.maximum(np.abs(want), 1e-3 * float(np.abs(want).max()))

    def compare(got):
        rel = np.abs(got - want) / scale
        return float(rel.max()), float((rel.max(axis=1) > 1e-4).mean())

    lat = phl.Lattice(r, reference_table=True)
    got = lat.filter(s, exact=True).cpu().numpy()
    res["gpu_bit_equal_to_reference"] = bool(lat.M == M_ref and np.array_equal(got.view(np.uint32), want.view(np.uint32)))
    res["M_gpu_reference_table"] = int(lat.M)
    mx, fr = compare(lat.filter(s).cpu().numpy())
    res["reference_table_default_arithmetic"] = {"max_rel": mx, "rows_beyond_1e-4": fr}
    lat.close()
    lat = phl.Lattice(r, reference_table=False)
    mx, fr = compare(lat.filter(s).cpu().numpy())
    res["clean_table_default_arithmetic"] = {"max_rel": mx, "rows_beyond_1e-4": fr, "M": int(lat.M)}
    lat.close()
    return res


def cpu_baseline(feat, H, W, L, d, check_gpu=True):
    """SURVEY.md 8(d) CPU baseline, timed on this box's host cores, beside (never inside) the GPU timing:
      (i)   1 thread on a bounded crop of the bench workload (the reference filters one image per thread);
      (ii)  batch mode: 8 independent items on P = min(8, nproc) worker PROCESSES, as the reference's
            `BatchedAdjacency(num_threads=8)` pool does (crf/gaussian_matrix.py:342,370-377);
      (iii) C1 (Tsukuba 384x288x16, BASELINE configs[0]) at FULL size, 1 thread.
    The lattice is rebuilt in every call (reference behaviour).  A reported baseline, not a target."""
    import subprocess
    import tempfile

    nproc = os.cpu_count() or 1
    ch, cw = min(H, 1024), min(W, 1536)      # ~10 s of single-thread CPU work at L=256
    ref = np.ascontiguousarray(feat[:ch, :cw].reshape(-1, d))
    rng = np.random.default_rng(4321)
    src = rng.random((ch * cw, L), dtype=np.float32)
    src /= src.sum(1, keepdims=True)
    kind, dt, M, stages, cpu_out = _cpu_filter_once(src, ref, want_out=True)
    out = {"value": round(ch * cw * L / dt / 1e6, 3), "unit": "Mpixel-labels/s", "cores": 1, "kind": kind,
           "sample": f"top-left {cw}x{ch} crop of the same features, L={L}, 1 thread, lattice rebuilt per call (reference behaviour)",
           "seconds": round(dt, 3), "M_over_n": round(M / (ch * cw), 4),
           "stage_seconds": {k: round(float(v), 4) for k, v in stages.items()},
           "nproc": nproc, "cpu_model": _cpu_model(),
           "compiler": "g++ -O2 (oracle/build_ref.sh)" if kind == "reference" else "gcc -O2 -ffp-contract=off (oracle/Makefile)"}

    if check_gpu:
        out.update(gpu_vs_reference(ref, src, cpu_out, M))
    del cpu_out

    # (ii) the reference's batch mode: one worker process per item, P at a time
    items, P = 8, min(8, nproc)
    bh, bw = min(H, 512), min(W, 768)        # 8 items of ~2.5 s each
    bref = np.ascontiguousarray(feat[:bh, :bw].reshape(-1, d))
    bsrc = np.ascontiguousarray(src[:bh * bw])
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "item.npz")
        np.savez(path, ref=bref, src=bsrc)
        env = dict(os.environ, OMP_NUM_THREADS="1")
        t0 = time.time()
        done, running, per_item = 0, [], []
        while done < items:
            while len(running) < P and done + len(running) < items:
                running.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path],
                                                stdout=subprocess.PIPE, env=env))
            proc = running.pop(0)
            o, _ = proc.communicate()
            if proc.returncode != 0:
                for q in running:
                    q.kill()
                raise RuntimeError("cpu_baseline: worker failed")
            per_item.append(json.loads(o.decode().strip().splitlines()[-1])["seconds"])
            done += 1
        wall = time.time() - t0
    out["batch"] = {"value": round(items * bh * bw * L / wall / 1e6, 3), "unit": "Mpixel-labels/s", "cores": P,
                    "processes": P, "items": items, "seconds": round(wall, 3),
                    "seconds_per_item_in_worker": round(float(np.mean(per_item)), 3),
                    "sample": f"{items} independent {bw}x{bh}x{L} items (crops of the same features), one fresh worker "
                              f"process per item, {P} at a time (mp.Pool(num_threads=8) semantics); wall time includes process start"}

    # (iii) C1 at full size
    H1, W1, L1, _ = WORKLOADS["c1"]
    f1 = synthetic_features(H1, W1).reshape(-1, d)
    s1 = rng.random((H1 * W1, L1), dtype=np.float32)
    s1 /= s1.sum(1, keepdims=True)
    reps, t = 5, []
    for _ in range(reps):
        _, dt1, M1, _ = _cpu_filter_once(s1, f1)[:4]
        t.append(dt1)
    out["c1_full"] = {"value": round(H1 * W1 * L1 / min(t) / 1e6, 3), "unit": "Mpixel-labels/s", "cores": 1,
                      "seconds": round(min(t), 4), "M_over_n": round(M1 / (H1 * W1), 4),
                      "sample": f"C1 {W1}x{H1}x{L1} full size (synthetic features of the SURVEY 8d recipe), best of {reps} calls, lattice rebuilt per call"}
    return out


# ---- launcher ----------------------------------------------------------------------------------
def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch(n, argv):
    """`python bench.py --gpus N` without a torchrun environment: start N fresh rank processes.

    This parent has not imported torch and never touches the GPU, so nothing that has initialised
    HIP is ever forked or re-executed.  Children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT (127.0.0.1 rendezvous) and HSA_ENABLE_IPC_MODE_LEGACY=0 in their environment
    before they start.  Rank 0's stdout (the ONE JSON line) is forwarded; the other ranks' stdout
    goes to stderr.  If a rank dies the others are terminated (by PID) and its code is returned."""
    import subprocess

    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0", PHL_BENCH_LAUNCHED="1")
    env.setdefault("OMP_NUM_THREADS", "4")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=None if r == 0 else sys.stderr))
    rc, alive = 0, list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:          # a failed rank leaves the others stuck in a collective
                    q.terminate()
    return rc


def dry_rank(args, rank, world):
    """--dry-launch: the launch / rendezvous / barrier / max-over-ranks / one-JSON-line plumbing over gloo on
    the CPU, without the filter (tests/test_bench_launch.py; there is no CPU filter to benchmark)."""
    import torch
    import torch.distributed as dist

    if os.environ.get("PHL_BENCH_TEST_FAIL_RANK") == str(rank):      # tests: a rank that dies before the rendezvous
        sys.exit(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0 + 1e-3 * rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    env = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "HSA_ENABLE_IPC_MODE_LEGACY")}
    seen = [None] * world
    dist.all_gather_object(seen, env)
    if rank == 0:
        print(json.dumps({"metric": "Mpixel-labels/s per CRF mean-field iter (splat+blur+slice)", "value": None,
                          "unit": "Mpixel-labels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "dry_launch": True, "backend": "gloo", "max_over_ranks_s": float(t.item()),
                          "launched_by": "bench.py launcher" if os.environ.get("PHL_BENCH_LAUNCHED") else "external (torchrun)",
                          "rank_env": seen}), file=JSON_OUT, flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    # (tools/step_seq.py: from an idle GPU the clocks take ~8 steps to come up; 3 warm-up steps left 1 % on the table)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--sigma-xy", type=float, default=SIGMA_XY, help="pixels per feature unit in x, y (SURVEY 8d: 8, 3, 30)")
    ap.add_argument("--sigma-c", type=float, default=SIGMA_C, help="colour scale of the synthetic features")
    ap.add_argument("--iid", action="store_true", help="SURVEY 8d stress case: iid U[0,1] colours (M/n -> d+1)")
    ap.add_argument("--tsukuba", default=None, metavar="SC,SP",
                    help="natural-image features: the stored Tsukuba frame upsampled to the workload size, "
                         "(rgb/SC, ij/diag/SP) as in DenseCrf.ipynb:142-146 (e.g. 0.1,0.1  0.08,0.03  0.125,0.01)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-regimes", action="store_true", help="skip the compact operating-point sweep (extra key `regimes`)")
    ap.add_argument("--clean-table", action="store_true",
                    help="build the lattice with one vertex per key instead of reproducing the reference's hash-table "
                         "behaviour across its doublings (duplicate vertices above M = 16383); see config.table")
    ap.add_argument("--no-tiles", action="store_true", help="plain gather kernels (A/B against the LDS-staged chunk path)")
    ap.add_argument("--exact", action="store_true", help="reference-exact arithmetic (bit-identical to the CPU path)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph (launch-bound sizes)")
    ap.add_argument("--force-rowtile", action="store_true", help="debug: run the row-band driver even with one rank")
    ap.add_argument("--mean-field", dest="mean_field", action="store_true", default=True,
                    help="also time one full mean-field iteration and its fused compatibility kernel (extra key; default on)")
    ap.add_argument("--no-mean-field", dest="mean_field", action="store_false")
    ap.add_argument("--no-small-image", action="store_true", help="skip the Tsukuba-sized extra (key c1_small_image)")
    ap.add_argument("--dry-launch", action="store_true", help="launcher / rendezvous plumbing only, gloo on CPU, no filter")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_worker:
        return cpu_worker(args.cpu_worker)

    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(launch(args.gpus, sys.argv[1:]))       # launcher: no torch import, no GPU touched in this process

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    # stdout carries the ONE JSON line and nothing else: native libraries (gloo, RCCL, rocBLAS) write their
    # chatter to fd 1, so fd 1 is pointed at stderr and the JSON goes to a private copy of the real stdout
    global JSON_OUT
    sys.stdout.flush()
    JSON_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.dry_launch:
        return dry_rank(args, rank, world)

    import torch

    if not torch.cuda.is_available():
        sys.exit("bench.py: no HIP device; the lattice filter has no CPU path to benchmark")
    ensure_built(local_rank)
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev      # ranks share a card only in rehearsals on a 1-GPU box (then: gloo, see below)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    backend = None
    if world > 1 or args.force_rowtile:
        import torch.distributed as dist

        if not (os.environ.get("MASTER_ADDR") and os.environ.get("MASTER_PORT")):
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        # one rank per GPU -> RCCL.  Two ranks on one card cannot form an RCCL communicator: gloo then
        # carries the (small) control traffic and the row-band payloads through host memory.
        backend = "nccl" if world <= ndev else "gloo"
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import phl

    H, W, L, desc = WORKLOADS[args.workload]
    d = 5
    tsu = tuple(float(x) for x in args.tsukuba.split(",")) if args.tsukuba else None
    feat, feat_desc = features_for(H, W, args.sigma_xy, args.sigma_c, args.iid, tsu)
    n_total = H * W

    rowtiled = world > 1 or args.force_rowtile
    if rowtiled and args.workload == "c5":
        rowtiled = False      # BASELINE configs[4]: independent volumes, one per GPU, no exchange
    if rowtiled:
        from phl import rowtile

        job = rowtile.RowTileFilter(feat, L, rank, world, device, dist)
        src = synthetic_values(torch, job.own_rows, W, L, job.row0, device)
        out_rt = torch.empty_like(src)
        step = lambda: job.filter(src, out=out_rt)
        build_ms, M, n_local = job.build_ms, job.M, job.n_local
        # which of the step's (bit-identical) schedules hides most of the REAL exchange is measured, not assumed
        # (RowTileFilter.autotune; collective); PHL_ROWTILE_MODE pins one instead
        if world > 1 and not os.environ.get("PHL_ROWTILE_MODE"):
            job.autotune(src, out_rt)
        extra = job.describe()
    else:
        ref = torch.from_numpy(feat.reshape(-1, d)).to(device)
        src = synthetic_values(torch, H, W, L, 1000 * rank, device)      # every rank its own volume
        torch.cuda.synchronize()
        t0 = time.time()
        ref_table = not args.clean_table
        lat = phl.Lattice(ref, reference_table=ref_table)
        torch.cuda.synchronize()
        build_ms = (time.time() - t0) * 1e3
        # steady-state rebuild (the reference builds a lattice in every filter call): build + destroy in a loop, so
        # that work arrays come from the cached scratch block and the lattice's own arrays from the block cache
        def warm_build(table):
            best = float("inf")
            for _ in range(4):
                t0 = time.time()
                lat_warm = phl.Lattice(ref, reference_table=table)
                torch.cuda.synchronize()
                best = min(best, (time.time() - t0) * 1e3)
                lat_warm.close()
            return best

        build_warm_ms = warm_build(ref_table)
        build_warm_other_ms = warm_build(not ref_table)
        lat.reserve(L)
        out = torch.empty_like(src)
        kw = dict(exact=args.exact, no_tiles=args.no_tiles)
        step = lambda: lat.filter(src, out=out, **kw)
        M, n_local = lat.M, n_total
        extra = {"tiles": lat.tile_stats(L), "lattice_build_warm_ms": round(build_warm_ms, 2),
                 ("lattice_build_warm_ms_reference_table" if args.clean_table else "lattice_build_warm_ms_clean_table"):
                     round(build_warm_other_ms, 2)}

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
        t = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    volumes = world if (world > 1 and not rowtiled) else 1     # independent volumes: one per rank
    value = volumes * n_total * L / (dt / args.steps) / 1e6

    # the same step on the lattice built the other way (clean <-> reference table): same kernels, a handful of
    # duplicate vertices more or fewer
    if rank == 0 and world == 1 and not rowtiled:
        lat_o = phl.Lattice(ref, reference_table=args.clean_table)
        for _ in range(3):
            lat_o.filter(src, out=out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lat_o.filter(src, out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        extra["ms_per_step_reference_table" if args.clean_table else "ms_per_step_clean_table"] = round(e0.elapsed_time(e1) / 10, 4)
        extra["M_reference_table" if args.clean_table else "M_clean_table"] = int(lat_o.M)
        lat_o.close()
        del lat_o

    # ---- per-kernel timing with HIP events on the launch stream (rank-local lattice) ---------
    roofline = None
    stage_ms = {}
    if rowtiled:
        # rank 0's own band (own pixels + ghost vertices): the same kernels, timed locally
        lat = job.band.eng
        out = torch.empty_like(src)
        kw = dict(exact=False, no_tiles=False)
        extra["tiles"] = lat.tile_stats(L)
    if rank == 0:
        ev = lambda: torch.cuda.Event(enable_timing=True)
        stage_ms = stage_times(torch, lat, src, out, kw, max(3, min(args.steps, 10)))
        blur_launches = (d + 2) // 2
        dom = max(stage_ms, key=stage_ms.get)
        ab = algorithmic_bytes(n_local, M, L, d)
        # what the blur launches really move when two axes share a pass (vertex array read + written once per
        # launch, composed neighbour ids 32 B per vertex per pair, 8 B for an odd last axis)
        npair = (d + 1) // 2
        blur_fused_bytes = blur_launches * 8 * M * L + npair * 32 * M + ((d + 1) % 2) * 8 * M
        achieved = ab[dom] / (stage_ms[dom] * 1e-3) / 1e9
        staged = extra["tiles"]["staged_splat"] and not (args.no_tiles or args.exact)
        staged_sl = extra["tiles"]["staged_slice"] and not args.no_tiles
        kname = {"splat": "k_splat_tiled+k_splat_reduce" if staged else "k_splat",
                 "blur": f"{(d + 1) // 2}*k_blur2" + ("+k_blur" if (d + 1) % 2 else ""),
                 "slice": "k_slice_tiled" if staged_sl else "k_slice"}
        # (the committed counters were taken on the default features with the reference-table lattice)
        default_feat = tsu is None and not args.iid and args.sigma_xy == SIGMA_XY and args.sigma_c == SIGMA_C
        traffic, traffic_src = pmc_traffic(kname[dom], args.workload) if (not rowtiled and default_feat) else (None, None)
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
        per_stage = {k: {"ms": round(stage_ms[k], 4), "launches_per_step": (blur_launches if k == "blur" else 1),
                         "algorithmic_GBps": round(ab[k] / (stage_ms[k] * 1e-3) / 1e9, 1)} for k in stage_ms}
        # blur: the 8(d) figure counts d+1 single-axis passes; the launches move less (two axes per pass)
        per_stage["blur"]["fused_pass_bytes"] = int(blur_fused_bytes)
        per_stage["blur"]["fused_pass_GBps"] = round(blur_fused_bytes / (stage_ms["blur"] * 1e-3) / 1e9, 1)
        roofline = {"bound": "hbm", "kernel": kname[dom],
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": int(ab[dom]), "avg_launch_ms": round(stage_ms[dom], 4),
                    "per_stage": per_stage,
                    # north-star wording: blur-pass READ bytes (d+1)*4*M*L against the HBM-read roofline
                    "measured_copy_GBps": round(copy_gbs, 1), "frac_of_measured_copy": round(achieved / copy_gbs, 4),
                    "blur_read_frac_of_peak": round((d + 1) * 4 * M * L / (stage_ms["blur"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if rowtiled:
            roofline["scope"] = f"rank 0's band only ({n_local} pixels, {M} vertices incl. ghosts); kernels as in the 1-GPU run"

    mean_field = None
    if args.mean_field and rank == 0 and not rowtiled:
        mean_field = mean_field_iteration(torch, phl, lat, src, L, device)
        if args.workload == "c3":
            mean_field["reference_label_count"] = reference_label_count(torch, device)
    if args.mean_field and rowtiled and L % 4 == 0:
        # the whole mean-field iteration on row bands: only W @ Q exchanges anything, the compatibility product and the
        # softmax are per pixel -- every rank on its own rows (collective: every rank runs it)
        from crf.crf_module import charbonneir, compatibility_matrix, mean_field_step

        labels = torch.arange(L, dtype=torch.float32, device=device)
        Mu_b = compatibility_matrix(lambda a, b: charbonneir(a, b, 3.0), labels)
        gen = torch.Generator(device=device)
        gen.manual_seed(99 + rank)
        E0_b = torch.rand(src.shape, generator=gen, device=device) * 10.0
        Qn = torch.empty_like(src)
        Wb = lambda U: job.filter(U, subtract_input=True)
        for _ in range(4):
            mean_field_step(E0_b, Wb, Mu_b, src, out=Qn)
        sync_all()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            mean_field_step(E0_b, Wb, Mu_b, src, out=Qn)
        sync_all()
        dtm = time.perf_counter() - t0
        if dist is not None:
            tm = torch.tensor([dtm], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dtm = float(tm.item())
        if rank == 0:
            mean_field = {"ms": round(dtm / reps * 1e3, 3), "Mpixel_labels_per_s": round(n_total * L / (dtm / reps) / 1e6, 1),
                          "what": "row bands: filter - Q with the boundary exchange, then (.)@Mu + E0 + softmax(-.) on each rank's own rows; max over ranks"}
        del E0_b, Qn

    backward = None
    if args.mean_field and rank == 0 and not rowtiled and not (args.exact or args.no_tiles):
        backward = backward_pass(torch, lat, src, ref, args.workload)

    regimes = None
    default_features = tsu is None and not args.iid and args.sigma_xy == SIGMA_XY and args.sigma_c == SIGMA_C
    if (rank == 0 and world == 1 and not rowtiled and not args.no_regimes and default_features
            and not (args.exact or args.no_tiles or args.graph)):
        regimes = regime_sweep(torch, phl, H, W, L, d, src, out, ms_per_step, M)

    # ---- N > 1: the run validates itself ------------------------------------------------------
    check = None
    if world > 1 or args.force_rowtile:
        check = self_check(torch, phl, dist, backend, rank, world, device, rowtiled, job if rowtiled else None,
                           lat if not rowtiled else None, feat, H, W, L, d, src, out_rt if rowtiled else out)

    small = None
    if rank == 0 and world == 1 and not rowtiled and args.mean_field and default_features and not args.no_small_image:
        small = small_image(torch, phl, device)

    cpu = None
    if rank == 0 and world == 1 and not rowtiled and not args.no_cpu_baseline:
        cpu = cpu_baseline(feat, H, W, L, d)

    if rank == 0:
        line = {
            "metric": "Mpixel-labels/s per CRF mean-field iter (splat+blur+slice)",
            "value": round(value, 1), "unit": "Mpixel-labels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if (world > 1 and rowtiled) else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "H": H, "W": W, "L": L, "d": d, "features": feat_desc,
                       "sigma_xy": None if tsu else args.sigma_xy, "sigma_c": tsu[0] if tsu else args.sigma_c, "n": n_total, "M": int(M), "M_over_n": round(M / n_local, 4),
                       "parallelism": ("single GPU" if world == 1 and not rowtiled else
                                       f"row bands x{world} + RCCL boundary-vertex exchange" if rowtiled else
                                       f"{world} independent volumes, one per GPU, no collective"),
                       "launch": "hip graph replay" if (args.graph and not rowtiled) else "eager",
                       "arithmetic": "reference-exact (bit-identical to the CPU path)" if args.exact else "default (fp32-rounding-equivalent, ~1e-7 rel)",
                       # "reference": vertices, duplicates included, are the reference's (its hash table files the key in
                       # flight at every doubling from a stale slot, permutohedral.h:59-62,101-103) -- with --exact the
                       # output is bit-identical to the reference engine at this size (cpu_baseline.gpu_bit_equal_to_reference
                       # checks it in this run); "clean": one vertex per key.  Row bands are cut out of the whole image's
                       # reference-table lattice (phl_sub_lattice): "reference" too, checked by `check` in the same run.
                       "table": (job.band.table if rowtiled else ("clean" if args.clean_table else "reference"))},
            "ranks": world, "backend": backend, "devices_visible": ndev,
            "launched_by": "bench.py launcher" if os.environ.get("PHL_BENCH_LAUNCHED") else ("torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else "direct"),
            "lattice_build_ms": round(build_ms, 2),
            # the reference rebuilds its lattice in every filter call: the same metric with a (steady-state, warm
            # scratch) build added to every step -- SURVEY 8d asks for both
            "value_rebuild_each_iter": round(volumes * n_total * L / ((ms_per_step + extra.get("lattice_build_warm_ms", build_ms)) * 1e-3) / 1e6, 1),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if mean_field is not None:
            line["mean_field_iteration"] = mean_field
        if backward is not None:
            line["backward"] = backward
        if regimes is not None:
            line["regimes"] = regimes
        if small is not None:
            line["c1_small_image"] = small
        if check is not None:
            line["check"] = check
        line.update(extra)
        print(json.dumps(line), file=JSON_OUT, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if check is not None and not check["ok"]:
        sys.exit(f"bench.py: multi-rank result check FAILED: {check}")


CHECK_TOL = 1e-4     # north_star: "within 1e-4 relative per pixel"


def self_check(torch, phl, dist, backend, rank, world, device, rowtiled, job, lat, feat, H, W, L, d, src, out):
    """An N > 1 line must not print a number nobody verified (the first RCCL run of this code happens on the
    driver's node).  Row bands: every rank filters the WHOLE seeded volume through one single-image lattice on its
    own GPU -- no communication involved -- and compares the rows of its band with what the row-band step just
    produced; the worst relative error over all ranks is reduced with MAX.  Independent volumes (c5): the rank's
    default (LDS-staged) result against the gather kernels in the reference's summation order -- another code
    path over the same lattice -- plus run-to-run bit equality.  Also times the exchange (rowtile.exchange_probe)."""
    def rel(got, want):
        scale = torch.clamp(want.abs(), min=1e-3 * float(want.abs().max()))
        return float(((got - want).abs() / scale).max())

    res = {"tolerance": CHECK_TOL}
    if rowtiled:
        probe = job.exchange_probe(src, out)
        mine = torch.tensor([job.row0, job.own_rows], dtype=torch.int64)
        bands = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        if world > 1:
            if backend == "nccl":
                g = [b.to(device) for b in bands]
                dist.all_gather(g, mine.to(device))
                bands = [b.cpu() for b in g]
            else:
                dist.all_gather(bands, mine)
        else:
            bands = [mine]
        full = torch.cat([synthetic_values(torch, int(r), W, L, int(r0), device) for r0, r in bands])
        ref_bands = job.band.table == "reference"
        whole = phl.Lattice(torch.from_numpy(feat.reshape(-1, d)).to(device), reference_table=ref_bands)
        want = whole.filter(full)
        a = job.row0 * W
        err = rel(out, want[a:a + job.n_local])
        res.update(what="row-band result vs a single-lattice filter of the whole volume on the same GPU ("
                        + ("reference table: the bands are cut out of it" if ref_bands else "defect-free table, like the bands") + ")", **probe)
        whole.close()
        del want
        # ... and how far the bands (one defect-free lattice per band) are from the REFERENCE's result, whose hash table
        # files the key in flight at each doubling from a stale slot (permutohedral.h:59-62,101-103): the single lattice
        # built with the reference's table, same volume, same GPU.  Reported, not part of `ok`: the rows beyond 1e-4 are the
        # ones the reference's duplicate vertices touch.
        whole_ref = phl.Lattice(torch.from_numpy(feat.reshape(-1, d)).to(device), reference_table=True)
        want_ref = whole_ref.filter(full)[a:a + job.n_local]
        scale = torch.clamp(want_ref.abs(), min=1e-3 * float(want_ref.abs().max()))
        relr = ((out - want_ref).abs() / scale).max(dim=1).values
        far = torch.tensor([float((relr > CHECK_TOL).sum()), float(relr.numel()), float(relr.max())], dtype=torch.float64,
                           device=device if backend == "nccl" else "cpu")
        if world > 1:
            mx = far[2:].clone()
            dist.all_reduce(far[:2], op=dist.ReduceOp.SUM)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            far[2] = mx[0]
        res["vs_reference_table"] = {"rows_beyond_1e-4": float(far[0] / far[1]), "max_rel": float(far[2]), "M_reference": int(whole_ref.M),
                                     "what": "band rows vs the single lattice built with the reference's table (its duplicate vertices)"}
        whole_ref.close()
        del want_ref, full
    else:
        a = lat.filter(src)
        b = lat.filter(src)
        repeat = bool(torch.equal(a, b))
        err = rel(a, lat.filter(src, exact=True))
        if not repeat:
            err = float("inf")
        res.update(what="default path vs gather kernels (reference summation order) on this rank's volume; run-to-run bit equality",
                   repeatable=repeat)
        # ... and, since both of those walk ONE data structure, a crop against the CPU checker (its own lattice, its own
        # arithmetic) when oracle/ travelled with the tree: exact mode must be bit-identical, the default within tolerance
        try:
            from oracle import phl_oracle as po

            po.build_oracle()
            ch, cw = min(H, 96), min(W, 128)
            idx = (torch.arange(ch, device=device)[:, None] * W + torch.arange(cw, device=device)[None, :]).flatten()
            ref_c = np.ascontiguousarray(feat[:ch, :cw].reshape(-1, d))
            src_c = src[idx].contiguous()
            want = torch.from_numpy(po.Oracle(ref_c, faithful_table=True).filter(src_c.cpu().numpy())).to(device)
            small = phl.Lattice(torch.from_numpy(ref_c).to(device), reference_table=True)
            bit = bool(torch.equal(small.filter(src_c, exact=True), want))
            e2 = rel(small.filter(src_c), want)
            small.close()
            res.update(oracle_crop=f"{cw}x{ch}x{L}", oracle_crop_exact_bit_equal=bit, oracle_crop_default_max_rel=e2)
            err = max(err, e2) if bit else float("inf")
        except (ImportError, OSError, RuntimeError) as e:
            res["oracle_crop"] = f"not run ({type(e).__name__}: oracle/ not usable on this box)"
    t = torch.tensor([err], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    res["check_max_rel"] = float(t.item())
    res["ok"] = bool(res["check_max_rel"] <= CHECK_TOL)
    return res


def regime_sweep(torch, phl, H, W, L, d, src, out, base_ms, base_M):
    """SURVEY 8(d) "synthetic inputs": the same volume filtered under other feature distributions -- sigma_xy 3
    and 30, and natural-image features (the stored Tsukuba frame upsampled, DenseCrf.ipynb:142-146 scaling) -- so
    that the headline is not one operating point's speed.  Compact: M/n, ms per step, algorithmic GB/s (8d byte
    counts) and that figure relative to the default features', plus per stage the HIP-event time and the 8(d) bytes of the
    stage / time / 8 TB/s (`per_stage.*.frac`; blur's byte count is what its (d+1)/2 two-axis launches move --
    roofline.per_stage.blur.fused_pass_bytes -- and with few vertices those rows come out of L2 / the Infinity Cache,
    so its fraction is not an HBM figure there).  The full sweep (both large workloads, stage times,
    gather-kernel comparison, iid stress case) is tools/regimes.py -> profiles/r03_regimes.json."""
    n = H * W
    alg = lambda M: sum(algorithmic_bytes(n, M, L, d).values())
    base_gbs = alg(base_M) / (base_ms * 1e-3) / 1e9
    rows = {"default": {"M_over_n": round(base_M / n, 4), "ms": round(base_ms, 4), "algorithmic_GBps": round(base_gbs, 1), "rel": 1.0}}
    for name, opt in (("sigma_xy=3", dict(sigma_xy=3.0)), ("sigma_xy=30", dict(sigma_xy=30.0)),
                      ("tsukuba 0.08/0.03", dict(tsukuba=(0.08, 0.03))), ("tsukuba 0.1/0.1", dict(tsukuba=(0.1, 0.1))),
                      ("(x, y, disparity) d=3", dict(xyd=1.0))):
        feat, _ = features_for(H, W, **opt)
        dd = feat.shape[-1]
        alg = lambda M, dd=dd: sum(algorithmic_bytes(n, M, L, dd).values())
        lat = phl.Lattice(torch.from_numpy(feat.reshape(-1, dd)).to(src.device))
        for _ in range(4):
            lat.filter(src, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            lat.filter(src, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gbs = alg(lat.M) / (ms * 1e-3) / 1e9
        st = lat.tile_stats(L)
        # the three stages on their own (HIP events on the launch stream): 8(d) bytes of the stage / its time / 8 TB/s
        stages = stage_times(torch, lat, src, out, {}, 3)
        sb = algorithmic_bytes(n, lat.M, L, dd)
        # blur: what its (d+1)/2 two-axis launches move (as roofline.per_stage.blur.fused_pass_bytes), not 8(d)'s d+1 passes
        sb["blur"] = ((dd + 2) // 2) * 8 * lat.M * L + ((dd + 1) // 2) * 32 * lat.M + ((dd + 1) % 2) * 8 * lat.M
        rows[name] = {"M_over_n": round(lat.M / n, 4), "ms": round(ms, 4), "Mpixel_labels_per_s": round(n * L / (ms * 1e-3) / 1e6, 1),
                      "algorithmic_GBps": round(gbs, 1), "rel": round(gbs / base_gbs, 3),
                      "per_stage": {k: {"ms": round(v, 4), "frac": round(sb[k] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} for k, v in stages.items()},
                      "max_local_vertices": st["max_local_vertices"], "staged": [st["staged_splat"], st["staged_slice"]]}
        lat.close()
        del lat
    return rows


def small_image(torch, phl, device):
    """BASELINE configs[0] geometry (Tsukuba 384x288x16, 5 mean-field iterations) -- the only size the reference's
    notebook runs (DenseCrf.ipynb:95-115) -- on the GPU, eagerly and replayed from one HIP graph: the two agree, i.e.
    at this size the chain of seven dependent few-microsecond kernels per filter is what takes the time, not the host
    issuing them.  cpu_baseline.c1_full is the reference engine's time for ONE filter call at this size."""
    import crf.crf_module as cm
    from crf.gaussian_matrix import LatticeGaussian

    H, W, L, _ = WORKLOADS["c1"]
    d, niters = 5, 5
    ref = torch.from_numpy(synthetic_features(H, W).reshape(-1, d)).to(device)
    E0 = torch.rand((H * W, L), device=device, generator=torch.Generator(device=device).manual_seed(7)) * 10.0
    labels = torch.arange(L, dtype=torch.float32, device=device)
    Mu = cm.compatibility_matrix(lambda a, b: cm.charbonneir(a, b, 3.0), labels)
    Wop = LatticeGaussian(ref)
    lat = phl.lattice_for(ref)
    src = torch.softmax(-E0, dim=1)
    out = torch.empty_like(src)

    def wall(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    res = {"workload": f"c1: {W}x{H}x{L}, d=5, {niters} mean-field iterations", "M_over_n": round(lat.M / (H * W), 4)}
    res["filter_ms_eager"] = round(wall(lambda: lat.filter(src, out=out), 200), 5)
    g = torch.cuda.CUDAGraph()
    lat.filter(src, out=out)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        lat.filter(src, out=out)
    res["filter_ms_graph"] = round(wall(g.replay, 200), 5)
    res["mean_field_ms_eager"] = round(wall(lambda: cm.mean_field_infer(E0, Wop, Mu, niters), 50), 4)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        cm.mean_field_infer(E0, Wop, Mu, niters)             # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=side):
        cm.mean_field_infer(E0, Wop, Mu, niters)
    res["mean_field_ms_graph"] = round(wall(g2.replay, 50), 4)
    res["Mpixel_labels_per_s_per_iter_graph"] = round(H * W * L * niters / (res["mean_field_ms_graph"] * 1e-3) / 1e6, 1)
    # the notebook's own call shape (DenseCrf.ipynb:142-152,173): E_0, ref and Mu are CPU tensors.  Staged once, iterated
    # on the device, Q copied back once (crf_module._mean_field_infer_staged): PCIe-inclusive wall time of the whole call.
    E0c, Muc, Wc = E0.cpu(), Mu.cpu(), LatticeGaussian(ref.cpu())
    res["mean_field_ms_cpu_tensors"] = round(wall(lambda: cm.mean_field_infer(E0c, Wc, Muc, niters), 20), 4)
    res["cpu_tensors_what"] = (f"E_0 [{H * W}, {L}] pageable host memory -> device, {niters} iterations on the "
                               "device, Q -> host (pinned); includes both PCIe crossings")
    return res


def backward_pass(torch, lat, src, ref, workload):
    """LatticeFilter.backward (crf/gaussian_matrix.py:435-468) through the fused kernels (phl_filter_grad): both gradients
    of sum(g * filter(src, ref)) on the bench workload, HIP events around back-to-back calls.  Extra key (the time used to
    be asserted inside the parity suite)."""
    import phl

    g = torch.randn(src.shape, device=src.device, generator=torch.Generator(device=src.device).manual_seed(3))
    try:
        for _ in range(2):
            lat.filter_grad(src, g, ref)
    except phl.PhlError as e:
        return {"unsupported": str(e)}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        lat.filter_grad(src, g, ref)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    n, L = src.shape
    return {"ms": round(ms, 3), "workload": workload, "what": "grad wrt source and features, two fused passes of wide splat -> blur -> contracting slice",
            "wide_operand_GB_the_reference_formulation_would_move": round(2 * n * 2 * L * (1 + ref.shape[1]) * 4 / 1e9, 1),
            "lattice_plus_workspaces_GB": round(lat.device_bytes / 1e9, 2)}


def reference_label_count(torch, device):
    """The reference takes max_disp = w // 6 labels (crf/depth.py:40): 231 at configs[1]'s 1390 columns -- not a multiple of 4,
    i.e. rows that are not made of 16-byte pieces.  mean_field_infer runs such counts padded with labels of probability
    exactly 0 (crf_module._label_pad); timed here next to the neighbouring multiple of 4.  Extra key."""
    import crf.crf_module as cm
    from crf.gaussian_matrix import LatticeGaussian

    H, W, _, _ = WORKLOADS["c2"]
    ref = torch.from_numpy(synthetic_features(H, W).reshape(-1, 5)).to(device)
    Wop = LatticeGaussian(ref)
    res = {"workload": f"{W}x{H}, d=5, 5 mean-field iterations (mean_field_infer on device tensors)"}
    for L in (W // 6, (W // 6 + 3) // 4 * 4):
        E0 = torch.rand((H * W, L), device=device, generator=torch.Generator(device=device).manual_seed(L)) * 10.0
        Mu = cm.compatibility_matrix(lambda a, b: cm.charbonneir(a, b, 3.0), torch.arange(L, dtype=torch.float32, device=device))
        for _ in range(2):
            cm.mean_field_infer(E0, Wop, Mu, 5)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            Q = cm.mean_field_infer(E0, Wop, Mu, 5)
        e1.record()
        torch.cuda.synchronize()
        assert Q.shape == (H * W, L)
        res[f"L={L}_ms"] = round(e0.elapsed_time(e1) / 3, 2)
        del E0, Q
    return res


def mean_field_iteration(torch, phl, lat, Q, L, device):
    """One full mean-field iteration of crf/crf_module.py:49-52 around the cached lattice:
    G = (filter(Q) - Q) @ Mu ; Q' = softmax(-(E0 + G)).  Extra key, not the headline metric."""
    from crf.crf_module import charbonneir, compatibility_matrix, mean_field_step

    labels = torch.arange(L, dtype=torch.float32, device=device)
    Mu = compatibility_matrix(lambda a, b: charbonneir(a, b, 3.0), labels)
    g = torch.Generator(device=device)
    g.manual_seed(99)
    E0 = torch.rand(Q.shape, generator=g, device=device) * 10.0
    Qn = torch.empty_like(Q)
    W = lambda U: lat.filter(U, subtract_input=True)
    for _ in range(6):                  # steady clocks (tools/compat_seq.py: ~25 ms of back-to-back work from idle)
        mean_field_step(E0, W, Mu, Q, out=Qn)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    e0, e1 = ev(), ev()
    reps = 10
    e0.record()
    for _ in range(reps):
        mean_field_step(E0, W, Mu, Q, out=Qn)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # the fused compatibility product + softmax alone, steady state (HIP events around `reps` back-to-back launches after
    # the warm-up above), in both arithmetic forms: "split" (the default for 176 < L <= 256: bf16 matrix cores on operands
    # split three ways, six exact partial products, f32 accumulation -- a streaming pass by its bytes) and "f32" (the
    # f32-input matrix cores, an fma chain in k order -- MFMA-bound); errors of both against float64 on a row sample
    X = W(Q)
    n = Q.shape[0]
    split_ok = 128 < L <= 256 and L % 4 == 0
    times = {}
    for arith in (("split", "f32") if split_ok else ("f32",)):
        for _ in range(4):
            phl.compat_softmax(E0, X, Mu, out=Qn, arith=arith)
        e2, e3 = ev(), ev()
        e2.record()
        for _ in range(reps):
            phl.compat_softmax(E0, X, Mu, out=Qn, arith=arith)
        e3.record()
        torch.cuda.synchronize()
        times[arith] = e2.elapsed_time(e3) / reps
    rows = torch.cat([torch.arange(0, 2048, device=device), torch.arange(n // 2, n // 2 + 2048, device=device), torch.arange(n - 2048, n, device=device)])
    want = -(E0[rows].double() + X[rows].double() @ Mu.double())          # the energies (logits epilogue): what the product's rounding shows in
    errs = {}
    for arith in times:
        got = phl.compat_softmax(E0, X, Mu, out=Qn, arith=arith, logits=True)
        errs[arith] = float((got[rows].double() - want).abs().max()) / float(want.abs().max())
    used = (os.environ.get("PHL_COMPAT_ARITH") or ("split" if L > 176 else "f32")) if split_ok else "f32"
    cms = times[used]
    F32_MFMA_PEAK = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak, TFLOP/s
    f32_tflops = 2.0 * n * L * L / (times["f32"] * 1e-3) / 1e12
    compat = {"kernel": "k_compat_split" if used == "split" else "k_compat_softmax", "arith": used, "ms": round(cms, 4),
              "hbm_bytes_algorithmic": int(3 * 4 * n * L), "algorithmic_GBps": round(3 * 4 * n * L / (cms * 1e-3) / 1e9, 1),
              "frac_of_8TBps": round(3 * 4 * n * L / (cms * 1e-3) / 8e12, 3),
              "f32_matrix_cores": {"kernel": "k_compat_softmax", "ms": round(times["f32"], 4), "tflops": round(f32_tflops, 1),
                                   "frac_of_157.3": round(f32_tflops / F32_MFMA_PEAK, 3)},
              "energy_err_vs_float64_over_max_energy_on_6144_rows": {k: float(f"{v:.3e}") for k, v in errs.items()},
              "what": "softmax(-(E0 + X@Mu)) in one kernel: reads E0 and X, writes Q; G and E never exist in HBM"}
    if used == "split":
        compat["bf16_tflops_six_products"] = round(6 * 2.0 * n * L * L / (cms * 1e-3) / 1e12, 1)
    return {"ms": round(ms, 3), "Mpixel_labels_per_s": round(n * L / (ms * 1e-3) / 1e6, 1),
            "what": "filter - Q (fused), (.)@Mu + E0 + softmax(-.)", "compat": compat}


if __name__ == "__main__":
    main()
