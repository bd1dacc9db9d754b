#!/usr/bin/env python3
"""Debug: tracking loss of the fused engine vs the reference-shaped path, pixel by pixel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.test_gpu_fused import _scene, _reference_grads
from splatam_amd import slam
from splatam_amd.fused import FusedEngine

aniso = len(sys.argv) > 1 and sys.argv[1] == "aniso"
params, variables, frame, cam = _scene(20000, 320, 240, aniso=aniso, seed=3)
cfg = slam.REPLICA_TRACKING
loss_ref, g_ref = _reference_grads(params, variables, frame, cfg, tracking=True)
eng = FusedEngine(params, cam)
eng.loss_backward(frame, 1, cfg, tracking=True)
torch.cuda.synchronize()
o = eng.buf['out6']
tg = slam.transform_to_frame(params, 1, False, False)
with torch.no_grad():
    rv = slam.transformed_params2rendervar(params, tg)
    im_ref, _, _ = slam.Renderer(raster_settings=cam)(**rv)
    dv = slam.transformed_params2depthplussilhouette(params, frame['w2c'], tg)
    ds_ref, _, _ = slam.Renderer(raster_settings=cam)(**dv)
ref6 = torch.cat([im_ref, ds_ref])
for ch in range(6):
    d = (o[ch] - ref6[ch]).abs()
    print(f"ch{ch}: max abs diff {float(d.max()):.3e}  n>1e-4: {int((d > 1e-4).sum())}  ref max {float(ref6[ch].abs().max()):.3f}")

def loss_terms(x6):
    depth, sil, dsq = x6[3:4], x6[4], x6[5:6]
    unc = dsq - depth ** 2
    nan_mask = (~torch.isnan(depth)) & (~torch.isnan(unc))
    mask = (frame['depth'] > 0) & nan_mask & (sil > cfg['sil_thres'])
    ld = torch.where(mask, (frame['depth'] - depth).abs(), torch.zeros_like(depth)).sum()
    li = torch.where(mask.expand(3, -1, -1), (frame['im'] - x6[0:3]).abs(), torch.zeros_like(x6[0:3])).sum()
    return mask, float(ld), float(li)
m_ref, ld_ref, li_ref = loss_terms(ref6)
m_eng, ld_eng, li_eng = loss_terms(o)
S = eng.buf['d_cam'].cpu().numpy()[8:12]
print("mask count ref/eng-render:", int(m_ref.sum()), int(m_eng.sum()), " mismatching pixels:", int((m_ref != m_eng).sum()))
print(f"depth L1: ref-render {ld_ref:.4f}  torch-on-engine-render {ld_eng:.4f}  kernel {S[0]:.4f}")
print(f"im    L1: ref-render {li_ref:.4f}  torch-on-engine-render {li_eng:.4f}  kernel {S[1]:.4f}")
print(f"loss: reference {loss_ref:.4f}  kernel {eng.loss():.4f}  recomputed {ld_eng + 0.5 * li_eng:.4f}")
sil = ref6[4]
for eps in (1e-7, 1e-6, 1e-5, 1e-4, 1e-3):
    print(f"pixels with |sil-0.99| < {eps:g}: {int(((sil - 0.99).abs() < eps).sum())}")
# gradient of the loss w.r.t. the 6 planes: kernel vs autograd on the engine's own render
x = o.detach().clone().requires_grad_(True)
depth, silx, dsq = x[3:4], x[4], x[5:6]
mask = m_eng
l = torch.where(mask, (frame['depth'] - depth).abs(), torch.zeros_like(depth)).sum() + 0.5 * torch.where(mask.expand(3, -1, -1), (frame['im'] - x[0:3]).abs(), torch.zeros_like(x[0:3])).sum()
l.backward()
gd = (x.grad - eng.buf['dL_dout6']).abs()
print("dL/dout6 kernel vs autograd: max diff", float(gd.max()), " n differing:", int((gd > 1e-6).sum()))
d = eng.buf['d_cam'].cpu().numpy()
print("pose grad engine:", d[:7])
print("pose grad ref   :", g_ref['cam_unnorm_rots'][0, :, 1].cpu().numpy(), g_ref['cam_trans'][0, :, 1].cpu().numpy())
