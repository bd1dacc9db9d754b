"""Loop-level recorder shared by tests/golden/make_golden_loop.py (which runs the REFERENCE's own ``rgbd_slam``,
/root/reference/scripts/splatam.py:455-990) and by the tests that hold ``splatam_amd.pipeline.rgbd_slam`` to that recording.

TEST INFRASTRUCTURE.  ``LoopRecorder`` wraps the names a SplaTAM loop resolves at call time in the module that holds the loop --
``get_loss``, ``initialize_optimizer`` (and the ``step`` of the optimizer it returns), ``add_new_gaussians``, ``prune_gaussians``,
``keyframe_selection_overlap``, ``initialize_camera_pose`` -- calls through to whatever was there, and writes down what the loop did:

    kind       a                b             c            d          value
    POSE0      time_idx         forward_prop
    OPT        tracking
    LOSS       iter_time_idx    tracking      mapping      do_ba      float(loss)
    STEP
    ADD        time_idx         rows before   rows after
    PRUNE      iter             rows before   rows after
    KFSEL      candidates       k             first index into ``selected``, count

The same recorder on the reference's module and on ``splatam_amd.slam`` / ``pipeline`` / ``plugin`` gives two call sequences that
must be EQUAL: keyframe cadence, the map_every / add_new_gaussians gating, the selected-keyframe window and the random view of every
mapping iteration, the prune schedule, the doubled tracking budget, the number of optimizer steps.
"""
from __future__ import annotations

import json

import numpy as np
import torch

POSE0, OPT, LOSS, STEP, ADD, PRUNE, KFSEL = range(7)
KIND_NAMES = ("POSE0", "OPT", "LOSS", "STEP", "ADD", "PRUNE", "KFSEL")


class LoopRecorder:
    def __init__(self, read_values=True):
        self.events, self.values, self.selected = [], [], []
        self.read_values = read_values
        self._undo = []

    # ------------------------------------------------------------------ plumbing
    def _emit(self, kind, a=0, b=0, c=0, d=0, value=float("nan")):
        self.events.append((kind, int(a), int(b), int(c), int(d)))
        self.values.append(float(value))

    def _patch(self, module, name, make):
        if not hasattr(module, name):
            return
        orig = getattr(module, name)
        setattr(module, name, make(orig))
        self._undo.append((module, name, orig))

    def restore(self):
        for module, name, orig in reversed(self._undo):
            setattr(module, name, orig)
        self._undo = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.restore()

    # ------------------------------------------------------------------ the wrapped names
    def wrap(self, module):
        """Wrap every loop name ``module`` has (the reference's scripts/splatam.py module, splatam_amd.slam, .pipeline, .plugin)."""
        rec = self

        def get_loss(orig):
            def f(params, curr_data, variables, iter_time_idx, *a, **k):
                out = orig(params, curr_data, variables, iter_time_idx, *a, **k)
                value = float(out[0].detach()) if rec.read_values else float("nan")
                rec._emit(LOSS, iter_time_idx, bool(k.get('tracking', False)), bool(k.get('mapping', False)), bool(k.get('do_ba', False)), value)
                return out
            return f

        def initialize_optimizer(orig):
            def f(params, lrs_dict, tracking):
                opt = orig(params, lrs_dict, tracking)
                rec._emit(OPT, bool(tracking))
                inner = opt.step

                def step(*a, **k):
                    rec._emit(STEP)
                    return inner(*a, **k)
                opt.step = step
                return opt
            return f

        def add_new_gaussians(orig):
            def f(params, variables, curr_data, sil_thres, time_idx, *a, **k):
                before = int(params['means3D'].shape[0])
                out = orig(params, variables, curr_data, sil_thres, time_idx, *a, **k)
                rec._emit(ADD, time_idx, before, int(out[0]['means3D'].shape[0]))
                return out
            return f

        def prune_gaussians(orig):
            def f(params, variables, optimizer, iter, prune_dict):
                before = int(params['means3D'].shape[0])
                out = orig(params, variables, optimizer, iter, prune_dict)
                rec._emit(PRUNE, iter, before, int(out[0]['means3D'].shape[0]))
                return out
            return f

        def keyframe_selection_overlap(orig):
            def f(gt_depth, w2c, intrinsics, keyframe_list, k, *a, **kw):
                out = orig(gt_depth, w2c, intrinsics, keyframe_list, k, *a, **kw)
                rec._emit(KFSEL, len(keyframe_list), k, len(rec.selected), len(out))
                rec.selected.extend(int(x) for x in out)
                return out
            return f

        def initialize_camera_pose(orig):
            def f(params, curr_time_idx, forward_prop):
                rec._emit(POSE0, curr_time_idx, bool(forward_prop))
                return orig(params, curr_time_idx, forward_prop)
            return f

        for name, make in (("get_loss", get_loss), ("initialize_optimizer", initialize_optimizer),
                           ("add_new_gaussians", add_new_gaussians), ("prune_gaussians", prune_gaussians),
                           ("keyframe_selection_overlap", keyframe_selection_overlap),
                           ("initialize_camera_pose", initialize_camera_pose)):
            self._patch(module, name, make)
        return self

    # ------------------------------------------------------------------ storage
    def arrays(self):
        return (np.asarray(self.events, dtype=np.int64).reshape(-1, 5), np.asarray(self.values, dtype=np.float64),
                np.asarray(self.selected, dtype=np.int64))


def describe(events, i):
    k, a, b, c, d = (int(x) for x in events[i])
    return f"#{i} {KIND_NAMES[k]}({a}, {b}, {c}, {d})"


def first_difference(ev_a, ev_b, ignore_row_counts=False):
    """Index and description of the first event where two recordings differ (None when equal).  ``ignore_row_counts``: the row
    counts of ADD / PRUNE events are not compared (a float32 rasterizer other than the recording's may put a pixel on the other
    side of the silhouette threshold; the counts are then compared with a tolerance by the caller)."""
    n = min(len(ev_a), len(ev_b))
    for i in range(n):
        x, y = [int(v) for v in ev_a[i]], [int(v) for v in ev_b[i]]
        if ignore_row_counts and x[0] == y[0] and x[0] in (ADD, PRUNE):
            x, y = x[:2], y[:2]
        if x != y:
            return i, f"{describe(ev_a, i)} != {describe(ev_b, i)}"
    if len(ev_a) != len(ev_b):
        return n, f"lengths {len(ev_a)} != {len(ev_b)}"
    return None


def per_frame_decisions(events, selected, keyframe_time_indices, num_frames, prune_dict):
    """The engine-independent DECISIONS of a recording, one dict per frame: how many tracking iterations ran, the map size after
    densification, the selected keyframes (indices into the keyframe list at that time), the view (time index) every mapping
    iteration rendered, the results of the prune calls that were on ``prune_dict``'s schedule, whether the frame became a keyframe,
    the map size when the frame was done."""
    kf = set(int(x) for x in keyframe_time_indices)
    frames = [dict(time_idx=t, tracking_iters=0, rows_after_add=None, selected=None, views=[], prunes=[], rows_end=None, keyframe=t in kf)
              for t in range(num_frames)]
    t, rows = 0, None
    for i in range(len(events)):
        k, a, b, c, d = (int(x) for x in events[i])
        if k == POSE0:
            frames[t]['rows_end'] = rows
            t = a
        elif k == LOSS and b:
            frames[t]['tracking_iters'] += 1
        elif k == LOSS and c:
            frames[t]['views'].append(a)
        elif k == ADD:
            for f in frames[:t]:                               # (a recording without prune calls first learns the row count here)
                if f['rows_end'] is None:
                    f['rows_end'] = b
            frames[t]['rows_after_add'] = rows = c
        elif k == PRUNE:
            on_schedule = a <= prune_dict['stop_after'] and a >= prune_dict['start_after'] and a % prune_dict['prune_every'] == 0
            assert on_schedule or b == c, describe(events, i)
            if on_schedule:
                frames[t]['prunes'].append((a, b, c))
            rows = c
        elif k == KFSEL:
            frames[t]['selected'] = [int(x) for x in selected[c:c + d]]
    frames[t]['rows_end'] = rows
    return frames


# ----------------------------------------------------------------------------------------------------------------------------------
# the recorded RGB-D sequence, as a dataset
# ----------------------------------------------------------------------------------------------------------------------------------

class RecordedRGBDSequence:
    """The frames a golden recording was made on, shaped like the reference's gradslam datasets:
    ``(color[H,W,3] in 0..255, depth[H,W,1], intrinsics[4,4], pose[4,4] camera-to-world relative to frame 0)``."""

    def __init__(self, gold, case, device="cpu"):
        self.color = torch.tensor(gold[f"{case}/frames/color"], device=device)
        self.depth = torch.tensor(gold[f"{case}/frames/depth"], device=device)
        self.k = torch.tensor(gold[f"{case}/frames/intrinsics"], device=device)
        self.pose = torch.tensor(gold[f"{case}/frames/poses"], device=device)

    def __len__(self):
        return self.color.shape[0]

    def __getitem__(self, t):
        return self.color[t], self.depth[t], self.k, self.pose[t]


def load_config(gold, case):
    return json.loads(str(gold[f"{case}/config"]))
