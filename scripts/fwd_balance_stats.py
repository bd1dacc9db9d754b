#!/usr/bin/env python3
"""CPU analysis for the forward composite: a wave walks the block lists of its quadrant's four 4x4 blocks in step, so a batch costs the
LONGEST of the four lists.  How many trips would remain if the tile's sixteen blocks were dealt to the sixteen (wave, 16-lane row) slots
by list length (longest four together, ...) instead of by position?  Geometry from the oracle's preprocess; alpha test as the composite
applies it (power <= 0, alpha >= 1/255); transmittance cut-off: `--sat` composites front to back and stops a pixel at T < 1e-4 as the
kernel does (a block's list then ends at its last unsaturated pixel's last entry).   usage: scripts/fwd_balance_stats.py [B|perpixel] [--sat]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import raster_ref as R          # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "B"
SAT = "--sat" in sys.argv
W, H, fx, fy, cx, cy = 1200, 680, 600.0, 600.0, 599.5, 339.5
t0 = time.time()
if which == "B":
    from splatam_amd import slam
    params, _ = slam.synthetic_params(300_000, W, H, fx, fy, cx, cy, num_frames=3, seed=0, device="cpu")
    means = params['means3D'].detach()
    scales = torch.exp(params['log_scales'].detach()).expand(-1, 3)
    rots = torch.nn.functional.normalize(params['unnorm_rotations'].detach())
    opac = torch.sigmoid(params['logit_opacities'].detach()).reshape(-1)
else:
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    z = 2.0 + 0.3 * torch.sin(xs / 90.0) * torch.cos(ys / 70.0)
    means = torch.stack([(xs - cx) / fx * z, (ys - cy) / fy * z, z], -1).reshape(-1, 3)
    scales = (z / (0.5 * (fx + fy))).reshape(-1, 1).expand(-1, 3).contiguous()
    rots = torch.tensor([[1.0, 0, 0, 0]]).expand(means.shape[0], -1).contiguous()
    opac = torch.full((means.shape[0],), 0.5)
cam = R.make_camera(W, H, fx, fy, cx, cy)
g = R.preprocess(means, None, scales, rots, None, cam)
vis = g.radii > 0
xy, conic, depth = g.xy.numpy(), g.conic.numpy(), g.depth.numpy()
rmin, rmax, op = g.rect_min.numpy(), g.rect_max.numpy(), opac.numpy()
gx, gy = (W + 15) // 16, (H + 15) // 16
idx = np.nonzero(vis.numpy())[0]
wx = (rmax[idx, 0] - rmin[idx, 0]); wy = (rmax[idx, 1] - rmin[idx, 1])
cnt = wx * wy
gi = np.repeat(idx, cnt)
off = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
tx = rmin[gi, 0] + off % np.repeat(wx, cnt); ty = rmin[gi, 1] + off // np.repeat(wx, cnt)
tile = ty * gx + tx
order = np.lexsort((gi, depth[gi], tile))
gi, tile, tx, ty = gi[order], tile[order], tx[order], ty[order]
print(f"{which}: instances {len(gi)} ({time.time() - t0:.1f} s)")
py_, px_ = np.meshgrid(np.arange(16, dtype=np.float32), np.arange(16, dtype=np.float32), indexing="ij")
bounds = np.flatnonzero(np.diff(tile)) + 1
starts = np.concatenate([[0], bounds]); ends = np.concatenate([bounds, [len(tile)]])
rng = np.random.default_rng(0)
sample = rng.choice(len(starts), size=min(len(starts), 600), replace=False)
tot = dict(now=0, sorted=0, ideal=0, visits=0, live=0, longest=0)
for ti in sample:
    a, b = starts[ti], ends[ti]
    k = gi[a:b]
    X = tx[a] * 16 + px_; Y = ty[a] * 16 + py_
    dx = xy[k, 0][:, None, None] - X[None]; dy = xy[k, 1][:, None, None] - Y[None]
    power = -0.5 * (conic[k, 0][:, None, None] * dx * dx + conic[k, 2][:, None, None] * dy * dy) - conic[k, 1][:, None, None] * dx * dy
    alpha = np.minimum(0.99, op[k][:, None, None] * np.exp(power))
    live = (power <= 0) & (alpha >= 1.0 / 255.0) & (X < W)[None] & (Y < H)[None]
    if SAT:
        T = np.ones((16, 16), np.float32); done = np.zeros((16, 16), bool)
        for e in range(len(k)):
            l = live[e] & ~done
            tt = T * (1 - alpha[e])
            stop = l & (tt < 1e-4)
            done |= stop
            upd = l & ~stop
            T = np.where(upd, tt, T)
            live[e] = l            # (a saturating pixel still takes the visit that stops it)
    # block b = (by, bx) 4x4; list length = entries with any live pixel in the block
    blk = live.reshape(len(k), 4, 4, 4, 4).any((2, 4))            # [e, by, bx]
    lens = blk.sum(0)                                             # [4, 4]
    # now: wave = quadrant (qy, qx), rows = its 2 x 2 blocks
    now = sum(int(lens[2 * qy:2 * qy + 2, 2 * qx:2 * qx + 2].max()) for qy in range(2) for qx in range(2))
    s = np.sort(lens.reshape(-1))[::-1]
    srt = int(s[0] + s[4] + s[8] + s[12])
    tot['now'] += now; tot['sorted'] += srt; tot['ideal'] += lens.sum() / 4.0; tot['visits'] += int(lens.sum()); tot['live'] += int(live.sum())
    tot['longest'] += int(lens.max()) * 4
print(f"tiles sampled {len(sample)}; block visits {tot['visits']}; live pixels per block visit {tot['live'] / tot['visits']:.2f} of 16")
print(f"trips (x 4 entries): by position {tot['now']}  by length {tot['sorted']} ({100 * tot['sorted'] / tot['now']:.1f} %)  perfectly even {tot['ideal']:.0f} "
      f"({100 * tot['ideal'] / tot['now']:.1f} %)   [every wave as long as the tile's longest list: {tot['longest']}]")
print(f"live lanes per trip: by position {100 * tot['live'] / (64 * tot['now']):.1f} %, by length {100 * tot['live'] / (64 * tot['sorted']):.1f} %  ({time.time() - t0:.1f} s)")

# ---- how tight is the composites' staging cull?  (box of {alpha >= 1/255} per 4x4 block + the radial test per 8x8 QUADRANT, render.hip gather())
# against the radial test per BLOCK and the exact per-pixel test
def kernel_cull(k, tx0, ty0, per_block_radial):
    a, b2, c = conic[k, 0], conic[k, 1], conic[k, 2]
    tau2 = 2.0 * np.log(255.0 * op[k])
    det = a * c - b2 * b2
    hx = np.sqrt(np.maximum(tau2, 0) * c / det) * 1.00001 + 0.01
    hy = np.sqrt(np.maximum(tau2, 0) * a / det) * 1.00001 + 0.01
    mid = 0.5 * (a + c)
    lam = mid - np.sqrt(np.maximum(0, mid * mid - det))
    mx, my = xy[k, 0], xy[k, 1]
    out = np.zeros((len(k), 4, 4), bool)
    for by in range(4):
        for bx in range(4):
            x0, y0 = tx0 + 4.0 * bx, ty0 + 4.0 * by
            box = (mx - hx <= x0 + 3) & (mx + hx >= x0) & (my - hy <= y0 + 3) & (my + hy >= y0)
            if per_block_radial:
                X0, Y0, ext = x0, y0, 3.0
            else:
                X0, Y0, ext = tx0 + 8.0 * (bx >> 1), ty0 + 8.0 * (by >> 1), 7.0
            ddx = np.maximum(np.maximum(X0 - mx, mx - (X0 + ext)), 0); ddy = np.maximum(np.maximum(Y0 - my, my - (Y0 + ext)), 0)
            rad = ~(lam * (ddx * ddx + ddy * ddy) > tau2 * 1.001 + 1e-3)
            out[:, by, bx] = box & rad & (tau2 >= 0)
    return out
tk = dict(kernel=0, block=0, exact=0, now_k=0, now_b=0)
for ti in sample[:300]:
    a, b = starts[ti], ends[ti]
    k = gi[a:b]
    X = tx[a] * 16 + px_; Y = ty[a] * 16 + py_
    dx = xy[k, 0][:, None, None] - X[None]; dy = xy[k, 1][:, None, None] - Y[None]
    power = -0.5 * (conic[k, 0][:, None, None] * dx * dx + conic[k, 2][:, None, None] * dy * dy) - conic[k, 1][:, None, None] * dx * dy
    alpha = np.minimum(0.99, op[k][:, None, None] * np.exp(power))
    live = (power <= 0) & (alpha >= 1.0 / 255.0)
    ex = live.reshape(len(k), 4, 4, 4, 4).any((2, 4))
    kc = kernel_cull(k, tx[a] * 16.0, ty[a] * 16.0, False); kb = kernel_cull(k, tx[a] * 16.0, ty[a] * 16.0, True)
    assert not (ex & ~kb).any(), "the per-block radial test must be conservative"
    tk['kernel'] += int(kc.sum()); tk['block'] += int(kb.sum()); tk['exact'] += int(ex.sum())
    for nm, m in (('now_k', kc), ('now_b', kb)):
        lens = m.sum(0)
        tk[nm] += sum(int(lens[2 * qy:2 * qy + 2, 2 * qx:2 * qx + 2].max()) for qy in range(2) for qx in range(2))
print(f"block visits over 300 tiles: staging cull of the kernels {tk['kernel']}, with the radial test per block {tk['block']} ({100 * tk['block'] / tk['kernel']:.1f} %), exact {tk['exact']} ({100 * tk['exact'] / tk['kernel']:.1f} %)")
print(f"trips (longest of a wave's four lists): {tk['now_k']} -> {tk['now_b']} ({100 * tk['now_b'] / tk['now_k']:.1f} %)")

# ---- the same question for the backward composite's quadrant lists (box + radial test per quadrant)
def quadrant_cull(k, tx0, ty0):
    a, b2, c = conic[k, 0], conic[k, 1], conic[k, 2]
    tau2 = 2.0 * np.log(255.0 * op[k])
    det = a * c - b2 * b2
    hx = np.sqrt(np.maximum(tau2, 0) * c / det) * 1.00001 + 0.01
    hy = np.sqrt(np.maximum(tau2, 0) * a / det) * 1.00001 + 0.01
    mid = 0.5 * (a + c)
    lam = mid - np.sqrt(np.maximum(0, mid * mid - det))
    mx, my = xy[k, 0], xy[k, 1]
    out = np.zeros((len(k), 2, 2), bool)
    for qy in range(2):
        for qx in range(2):
            x0, y0 = tx0 + 8.0 * qx, ty0 + 8.0 * qy
            box = (mx - hx <= x0 + 7) & (mx + hx >= x0) & (my - hy <= y0 + 7) & (my + hy >= y0)
            ddx = np.maximum(np.maximum(x0 - mx, mx - (x0 + 7)), 0); ddy = np.maximum(np.maximum(y0 - my, my - (y0 + 7)), 0)
            out[:, qy, qx] = box & ~(lam * (ddx * ddx + ddy * ddy) > tau2 * 1.001 + 1e-3) & (tau2 >= 0)
    return out
qk = qe = 0
for ti in sample[:300]:
    a, b = starts[ti], ends[ti]
    k = gi[a:b]
    X = tx[a] * 16 + px_; Y = ty[a] * 16 + py_
    dx = xy[k, 0][:, None, None] - X[None]; dy = xy[k, 1][:, None, None] - Y[None]
    power = -0.5 * (conic[k, 0][:, None, None] * dx * dx + conic[k, 2][:, None, None] * dy * dy) - conic[k, 1][:, None, None] * dx * dy
    alpha = np.minimum(0.99, op[k][:, None, None] * np.exp(power))
    live = (power <= 0) & (alpha >= 1.0 / 255.0)
    qe += int(live.reshape(len(k), 2, 8, 2, 8).any((2, 4)).sum()); qk += int(quadrant_cull(k, tx[a] * 16.0, ty[a] * 16.0).sum())
print(f"quadrant visits over 300 tiles: staging cull {qk}, exact {qe} ({100 * qe / qk:.1f} %)")

# ---- list entries that no quadrant of their tile visits (the tile rectangle comes from ceil(3 sigma), the live region from alpha >= 1/255)
dead_k = dead_e = tot_e = 0
for ti in sample[:300]:
    a, b = starts[ti], ends[ti]
    k = gi[a:b]
    X = tx[a] * 16 + px_; Y = ty[a] * 16 + py_
    dx = xy[k, 0][:, None, None] - X[None]; dy = xy[k, 1][:, None, None] - Y[None]
    power = -0.5 * (conic[k, 0][:, None, None] * dx * dx + conic[k, 2][:, None, None] * dy * dy) - conic[k, 1][:, None, None] * dx * dy
    alpha = np.minimum(0.99, op[k][:, None, None] * np.exp(power))
    live = (power <= 0) & (alpha >= 1.0 / 255.0)
    tot_e += len(k); dead_e += int((~live.any((1, 2))).sum()); dead_k += int((~quadrant_cull(k, tx[a] * 16.0, ty[a] * 16.0).any((1, 2))).sum())
print(f"list entries over 300 tiles: {tot_e}; visited by no quadrant: staging cull {dead_k} ({100 * dead_k / tot_e:.1f} %), exact {dead_e} ({100 * dead_e / tot_e:.1f} %)")

# ---- ... and how many of them a TILE-level test at binning time would drop: live box vs the tile's pixel centres, then + the radial test per tile
def tile_cull(k, tx0, ty0, radial):
    a, b2, c = conic[k, 0], conic[k, 1], conic[k, 2]
    tau2 = 2.0 * np.log(255.0 * op[k])
    det = a * c - b2 * b2
    hx = np.sqrt(np.maximum(tau2, 0) * c / det) * 1.00001 + 0.01
    hy = np.sqrt(np.maximum(tau2, 0) * a / det) * 1.00001 + 0.01
    mid = 0.5 * (a + c)
    lam = mid - np.sqrt(np.maximum(0, mid * mid - det))
    mx, my = xy[k, 0], xy[k, 1]
    keep = (mx - hx <= tx0 + 15) & (mx + hx >= tx0) & (my - hy <= ty0 + 15) & (my + hy >= ty0) & (tau2 >= 0)
    if radial:
        ddx = np.maximum(np.maximum(tx0 - mx, mx - (tx0 + 15)), 0); ddy = np.maximum(np.maximum(ty0 - my, my - (ty0 + 15)), 0)
        keep &= ~(lam * (ddx * ddx + ddy * ddy) > tau2 * 1.001 + 1e-3)
    return keep
db = dr = 0
for ti in sample[:300]:
    a, b = starts[ti], ends[ti]
    k = gi[a:b]
    db += int((~tile_cull(k, tx[a] * 16.0, ty[a] * 16.0, False)).sum()); dr += int((~tile_cull(k, tx[a] * 16.0, ty[a] * 16.0, True)).sum())
print(f"dropped by a tile-level box test {db} ({100 * db / tot_e:.1f} %), box + radial {dr} ({100 * dr / tot_e:.1f} %)")

# ---- the forward composite stages the list 255 entries at a time: the longest-of-four rule applies per BATCH
def batch_trips(which_cull):
    per_batch = whole = 0
    for ti in sample[:300]:
        a, b = starts[ti], ends[ti]
        k = gi[a:b]
        keep = tile_cull(k, tx[a] * 16.0, ty[a] * 16.0, False)          # what group binning files
        k = k[keep]
        m = kernel_cull(k, tx[a] * 16.0, ty[a] * 16.0, which_cull)      # [e, by, bx]
        for s in range(0, len(k), 255):
            lens = m[s:s + 255].sum(0)
            per_batch += sum(int(np.ceil(lens[2 * qy:2 * qy + 2, 2 * qx:2 * qx + 2].max() / 4.0)) for qy in range(2) for qx in range(2))
        lens = m.sum(0)
        whole += sum(int(np.ceil(lens[2 * qy:2 * qy + 2, 2 * qx:2 * qx + 2].max() / 4.0)) for qy in range(2) for qx in range(2))
    return per_batch, whole
pb, wh = batch_trips(True)
print(f"forward trips of four entries over 300 tiles, per 255-entry batch {pb}, if the whole list were one batch {wh} ({100 * wh / pb:.1f} %)")

# ---- ... and what dealing the sixteen blocks to the sixteen (wave, row) slots BY LENGTH, batch by batch, would leave
def batch_trips_dealt():
    dealt = 0
    for ti in sample[:300]:
        a, b = starts[ti], ends[ti]
        k = gi[a:b]
        k = k[tile_cull(k, tx[a] * 16.0, ty[a] * 16.0, False)]
        m = kernel_cull(k, tx[a] * 16.0, ty[a] * 16.0, True)
        for s in range(0, len(k), 255):
            l = np.sort(m[s:s + 255].sum(0).reshape(-1))[::-1]
            dealt += sum(int(np.ceil(l[4 * w] / 4.0)) for w in range(4))
    return dealt
dl = batch_trips_dealt()
print(f"blocks dealt by length per batch: {dl} trips ({100 * dl / pb:.1f} % of today's)")

# ---- ... with ONE dealing per tile, made from the first batch's lengths and kept for the tile's later batches (no pixel state moves)
def batch_trips_dealt_once(by_total=False):
    dealt = 0
    for ti in sample[:300]:
        a, b = starts[ti], ends[ti]
        k = gi[a:b]
        k = k[tile_cull(k, tx[a] * 16.0, ty[a] * 16.0, False)]
        m = kernel_cull(k, tx[a] * 16.0, ty[a] * 16.0, True).reshape(len(k), 16)
        first = m[:255].sum(0) if not by_total else m.sum(0)
        order = np.argsort(-first, kind="stable")
        for s in range(0, len(k), 255):
            l = m[s:s + 255].sum(0)[order]
            dealt += sum(int(np.ceil(l[4 * w:4 * w + 4].max() / 4.0)) for w in range(4))
    return dealt
d1 = batch_trips_dealt_once()
print(f"blocks dealt ONCE per tile by the first batch's lengths: {d1} trips ({100 * d1 / pb:.1f} % of today's)")
