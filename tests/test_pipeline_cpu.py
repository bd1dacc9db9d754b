"""Host logic of the frame loop (splatam_amd/pipeline.py), CPU only: keyframe selection against vectors produced by the
reference's own utils/keyframe_selection.py under the same seeds (tests/golden/make_golden_mapedit.py), the params.npz
round trip, and the config values the loop reads."""
import os

import numpy as np
import pytest
import torch

from splatam_amd import pipeline, slam

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mapedit_reference.npz"))


@pytest.mark.parametrize("k", [3, 10])
def test_keyframe_selection_matches_reference(k):
    depth, intr, w2c = torch.tensor(GOLD["kfsel/depth"]), torch.tensor(GOLD["kfsel/intrinsics"]), torch.tensor(GOLD["kfsel/w2c"])
    kfs = [{'id': 5 * i, 'est_w2c': torch.tensor(m)} for i, m in enumerate(GOLD["kfsel/est_w2c"])]
    torch.manual_seed(5)
    np.random.seed(5)
    sel = pipeline.keyframe_selection_overlap(depth, w2c, intr, kfs, k)
    assert [int(x) for x in sel] == GOLD[f"kfsel/selected_k{k}"].tolist()


def test_keyframe_selection_without_keyframes():
    depth, intr, w2c = torch.tensor(GOLD["kfsel/depth"]), torch.tensor(GOLD["kfsel/intrinsics"]), torch.tensor(GOLD["kfsel/w2c"])
    assert pipeline.keyframe_selection_overlap(depth, w2c, intr, [], 3) == GOLD["kfsel/selected_empty"].tolist() == []


def test_params_npz_round_trip(tmp_path):
    params, _ = slam.synthetic_params(50, 32, 24, 30.0, 30.0, 15.5, 11.5, num_frames=3, seed=1, device="cpu")
    path = pipeline.save_params(params, str(tmp_path))
    assert os.path.basename(path) == "params.npz"
    assert os.path.basename(pipeline.save_params(params, str(tmp_path), time_idx=7)) == "params7.npz"
    back = pipeline.load_params(path, device="cpu")
    assert set(back) == set(params)
    for k in params:
        assert torch.equal(back[k].detach(), params[k].detach()) and back[k].requires_grad


def test_replica_config_values():
    c = pipeline.replica_config()
    assert (c['map_every'], c['keyframe_every'], c['mapping_window_size']) == (1, 5, 24)
    assert c['tracking']['num_iters'] == 40 and c['mapping']['num_iters'] == 60
    assert c['tracking']['lrs']['cam_trans'] == 0.002 and c['mapping']['lrs']['logit_opacities'] == 0.05
    assert c['mapping']['pruning_dict']['prune_every'] == 20 and c['mapping']['sil_thres'] == 0.5


def test_matrix_to_quaternion():
    q = torch.nn.functional.normalize(torch.tensor([[0.9, 0.1, -0.3, 0.2]]))
    R = slam.build_rotation(q)[0]
    got = pipeline._matrix_to_quaternion(R)
    assert torch.allclose(got, q, atol=1e-6) or torch.allclose(got, -q, atol=1e-6)


def test_save_ply_layout(tmp_path):
    """splat.ply as /root/reference/scripts/export_ply.py:20-42 lays it out (17 float32 vertex properties; written without plyfile)."""
    from splatam_amd import export
    n = 37
    g = np.random.default_rng(0)
    means, rots, rgb = g.normal(size=(n, 3)), g.normal(size=(n, 4)), g.uniform(size=(n, 3))
    logit, log_s = g.normal(size=(n, 1)), g.normal(size=(n, 1))
    path = export.save_ply(str(tmp_path / "splat.ply"), means, log_s, rots, rgb, logit)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
    assert [ln.split()[-1] for ln in lines[3:]] == list(export.PLY_ATTRS) and all(ln.startswith("property float ") for ln in lines[3:])
    t = np.frombuffer(body, dtype="<f4").reshape(n, 17)
    np.testing.assert_allclose(t[:, 0:3], means.astype(np.float32))
    assert not t[:, 3:6].any()
    np.testing.assert_allclose(t[:, 6:9] * 0.28209479177387814 + 0.5, rgb, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(t[:, 9], logit[:, 0].astype(np.float32))
    np.testing.assert_allclose(t[:, 10:13], np.tile(log_s, (1, 3)).astype(np.float32))
    np.testing.assert_allclose(t[:, 13:17], rots.astype(np.float32))
    params = {'means3D': torch.tensor(means), 'log_scales': torch.tensor(log_s), 'unnorm_rotations': torch.tensor(rots),
              'rgb_colors': torch.tensor(rgb), 'logit_opacities': torch.tensor(logit)}
    assert open(export.export_params_ply(params, str(tmp_path / "b.ply")), "rb").read() == raw
