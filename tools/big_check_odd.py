"""Rows off the 16-byte grid at a size past 2^31 elements: 12.6 M pixels x 231 channels (the reference's w // 6 at 1390 columns)
through phl_filter's staging (k_copy_rows, 64-bit offsets) -- against the same tensor padded to 232 columns by hand."""
import sys
sys.path.insert(0, 'depth-estimation_amd'); sys.path.insert(0, '.')
import torch, phl, bench
H, W, L = 3072, 4096, 231
dev = torch.device('cuda')
lat = phl.Lattice(torch.from_numpy(bench.synthetic_features(H, W).reshape(-1, 5)).to(dev))
x = torch.rand((H * W, L), device=dev)
y = lat.filter(x)
xp = torch.zeros((H * W, L + 1), device=dev)
xp[:, :L] = x
yp = lat.filter(xp)
same = bool(torch.equal(y, yp[:, :L]))
err = float((y - yp[:, :L]).abs().max())
print('elements', H * W * L, 'odd-width filter equals the hand-padded one bitwise:', same, 'max abs diff', err, 'finite', bool(torch.isfinite(y).all()))
ys = lat.filter(x, subtract_input=True)
print('fused subtraction consistent:', float((ys - (y - x)).abs().max()))
ok = same and float((ys - (y - x)).abs().max()) <= 1e-5
print('BIG ODD CHECK', 'ok' if ok else 'FAILED')
sys.exit(0 if ok else 1)
