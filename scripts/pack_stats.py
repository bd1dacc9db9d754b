#!/usr/bin/env python3
"""CPU analysis for a packed backward composite: how many (Gaussian, 8x8 quadrant) visits of the backward pass touch only some of the
quadrant's four 4x4 blocks, and how many trips remain if visits whose block sets are disjoint share a trip (greedy, in depth order, so
no pixel sees its Gaussians out of order).  Geometry from the oracle's preprocess (torch, CPU); alpha test as the composite applies it
(power <= 0, alpha >= 1/255); transmittance cut-off ignored.   usage: scripts/pack_stats.py [B|perpixel] [seed]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import raster_ref as R          # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "B"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W, H, fx, fy, cx, cy = 1200, 680, 600.0, 600.0, 599.5, 339.5
t0 = time.time()
if which == "B":
    from splatam_amd import slam
    params, _ = slam.synthetic_params(300_000, W, H, fx, fy, cx, cy, num_frames=3, seed=seed, device="cpu")
    means = params['means3D'].detach()
    scales = torch.exp(params['log_scales'].detach()).expand(-1, 3) if params['log_scales'].shape[1] == 1 else torch.exp(params['log_scales'].detach())
    rots = torch.nn.functional.normalize(params['unnorm_rotations'].detach())
    opac = torch.sigmoid(params['logit_opacities'].detach()).reshape(-1)
else:   # the map initialize_first_timestep builds: one Gaussian per pixel, scale = depth / focal (1 px), opacity 0.5
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    z = 2.0 + 0.3 * torch.sin(xs / 90.0) * torch.cos(ys / 70.0)
    means = torch.stack([(xs - cx) / fx * z, (ys - cy) / fy * z, z], -1).reshape(-1, 3)
    scales = (z / (0.5 * (fx + fy))).reshape(-1, 1).expand(-1, 3).contiguous()
    rots = torch.tensor([[1.0, 0, 0, 0]]).expand(means.shape[0], -1).contiguous()
    opac = torch.full((means.shape[0],), 0.5)
cam = R.make_camera(W, H, fx, fy, cx, cy)
g = R.preprocess(means, None, scales, rots, None, cam)
vis = g.radii > 0
print(f"{which}: {int(vis.sum())} visible Gaussians of {means.shape[0]}  ({time.time() - t0:.1f} s)")
xy, conic, depth = g.xy.numpy(), g.conic.numpy(), g.depth.numpy()
rmin, rmax, op = g.rect_min.numpy(), g.rect_max.numpy(), opac.numpy()
gx = (W + 15) // 16
# instances (Gaussian, tile)
idx = np.nonzero(vis.numpy())[0]
wx = (rmax[idx, 0] - rmin[idx, 0]); wy = (rmax[idx, 1] - rmin[idx, 1])
cnt = wx * wy
gi = np.repeat(idx, cnt)
off = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
tx = rmin[gi, 0] + off % np.repeat(wx, cnt); ty = rmin[gi, 1] + off // np.repeat(wx, cnt)
tile = ty * gx + tx
order = np.lexsort((depth[gi], tile))
gi, tile, tx, ty = gi[order], tile[order], tx[order], ty[order]
print(f"instances {len(gi)}")
# per instance: live mask over the tile's 16 x 16 pixels -> per quadrant 4-bit block mask and live-pixel count
py_, px_ = np.meshgrid(np.arange(16, dtype=np.float32), np.arange(16, dtype=np.float32), indexing="ij")
blockmask = np.zeros((len(gi), 4), np.uint8); livepx = np.zeros((len(gi), 4), np.int32)
CH = 40000
for s in range(0, len(gi), CH):
    e = min(len(gi), s + CH); k = gi[s:e]
    dx = xy[k, 0][:, None, None] - (tx[s:e, None, None] * 16 + px_[None]); dy = xy[k, 1][:, None, None] - (ty[s:e, None, None] * 16 + py_[None])
    power = -0.5 * (conic[k, 0][:, None, None] * dx * dx + conic[k, 2][:, None, None] * dy * dy) - conic[k, 1][:, None, None] * dx * dy
    alpha = np.minimum(0.99, op[k][:, None, None] * np.exp(power))
    inside = ((tx[s:e, None, None] * 16 + px_[None]) < W) & ((ty[s:e, None, None] * 16 + py_[None]) < H)
    live = (power <= 0) & (alpha >= 1.0 / 255.0) & inside
    for q in range(4):
        qy, qx = (q >> 1) * 8, (q & 1) * 8
        sub = live[:, qy:qy + 8, qx:qx + 8]
        livepx[s:e, q] = sub.sum((1, 2))
        m = np.zeros(e - s, np.uint8)
        for b in range(4):
            by, bx = (b >> 1) * 4, (b & 1) * 4
            m |= (sub[:, by:by + 4, bx:bx + 4].any((1, 2)).astype(np.uint8) << b)
        blockmask[s:e, q] = m
print(f"evaluated ({time.time() - t0:.1f} s)")
visits = blockmask > 0
nv = int(visits.sum())
pop = np.array([bin(i).count("1") for i in range(16)])
print(f"visits (Gaussian, quadrant) {nv}; live pixels per visit {livepx[visits].mean():.1f} of 64 ({100 * livepx[visits].mean() / 64:.1f} %)")
hist = np.bincount(pop[blockmask[visits]], minlength=5)
print("blocks touched per visit: " + ", ".join(f"{b}: {100 * hist[b] / nv:.1f} %" for b in range(1, 5)))
print(f"block-granular lane use if each block were its own 16-lane trip: {100 * livepx[visits].sum() / (16 * pop[blockmask[visits]].sum()):.1f} % live")
# greedy in-order packing per (tile, quadrant)
trips = 0
bounds = np.flatnonzero(np.diff(tile)) + 1
starts = np.concatenate([[0], bounds]); ends = np.concatenate([bounds, [len(tile)]])
for q in range(4):
    col = blockmask[:, q]
    for a, b in zip(starts, ends):
        occ = 0
        for m in col[a:b]:
            if m == 0:
                continue
            if occ & m or occ == 0:
                trips += 1 if (occ & m or occ == 0) else 0
                occ = int(m)
            else:
                occ |= int(m)
print(f"trips after greedy in-order packing {trips} = {100 * trips / nv:.1f} % of the visits ({time.time() - t0:.1f} s)")
# windowed block rows: the quadrant's visits are taken K at a time (K pair-buffer slots); within a window each 16-lane row walks ITS block's
# entries, so a window costs max over the four blocks of its entries (trips), against K trips now
for K in (8, 16, 32, 64):
    tw = 0
    for q in range(4):
        col = blockmask[:, q]
        for a, b in zip(starts, ends):
            m = col[a:b]
            m = m[m > 0]
            if not len(m):
                continue
            pad = (-len(m)) % K
            mm = np.concatenate([m, np.zeros(pad, np.uint8)]).reshape(-1, K)
            per_block = np.stack([((mm >> bb) & 1).sum(1) for bb in range(4)], 1)
            tw += int(per_block.max(1).sum())
    print(f"windowed block rows, K = {K}: {tw} trips = {100 * tw / nv:.1f} % of the visits ({time.time() - t0:.1f} s)")
