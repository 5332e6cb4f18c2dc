"""How much would an LDS-staged blur save?  For chunks of W consecutive vertices (internal locality order) count the
distinct rows their 9-row stencils touch, per axis pair.  python tools/blur_reuse.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, phl
H, W_, L, _ = bench.WORKLOADS["c3"]
lat = phl.Lattice(torch.from_numpy(bench.synthetic_features(H, W_).reshape(-1, 5)).cuda())
M = lat.M
nbr_ft = lat.neighbors()                         # [d+1][M][2] first-touch ids
rows = lat.vertex_rows().cpu().numpy()           # row of ft vertex
ft_of_row = np.empty(M, np.int64); ft_of_row[rows] = np.arange(M)
def to_rows(x):                                  # ft id (or -1) -> row (or -1)
    return np.where(x >= 0, rows[np.maximum(x, 0)], -1)
nb = np.stack([to_rows(nbr_ft[a][ft_of_row]) for a in range(6)])      # [axis][row][2] in row numbering
for pair in range(3):
    a, b = 2 * pair, 2 * pair + 1
    v = np.arange(M)
    cols = [v]
    for s in (0, 1):
        cols.append(nb[a][:, s])
        bs = nb[b][:, s]
        cols.append(bs)
        for s2 in (0, 1):
            cols.append(np.where(bs >= 0, nb[a][np.maximum(bs, 0), s2], -1))
    st = np.stack(cols, 1)                       # [M][9]
    present = (st >= 0).sum() / M
    dist = np.abs(st[:, 1:] - v[:, None])[st[:, 1:] >= 0]
    print(f"pair {pair}: |row(neighbour) - row(vertex)|: median {np.median(dist):.0f}, 90th pct {np.percentile(dist, 90):.0f}, "
          f"within 512: {(dist <= 512).mean():.2%}, within 2048: {(dist <= 2048).mean():.2%}, within 4096: {(dist <= 4096).mean():.2%}")
    for Wc in (32, 64, 128, 256):
        nchunk = M // Wc
        u = 0
        span = 0
        for c in range(0, nchunk, max(1, nchunk // 400)):      # sample 400 chunks
            blk = st[c * Wc:(c + 1) * Wc].ravel()
            blk = blk[blk >= 0]
            uu = np.unique(blk)
            u += len(uu) / Wc
            span += (uu.max() - uu.min()) / Wc
        k = len(range(0, nchunk, max(1, nchunk // 400)))
        print(f"pair {pair}: rows present per vertex {present:.2f}; W={Wc:4d}: distinct rows per vertex {u / k:.2f}, id span / W {span / k:.1f}")
