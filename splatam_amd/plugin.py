"""Run-time plug-in: the fused iteration under the reference's OWN loop statements.

``scripts/splatam.py``'s tracking and mapping loops (/root/reference/scripts/splatam.py:690-711, :846-869) are

    loss, variables, losses = get_loss(params, data, variables, iter_time_idx, ...)
    loss.backward()
    [prune_gaussians / densify between backward and step, mapping only]
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)

with ``optimizer = initialize_optimizer(params, lrs, tracking=...)`` created per frame and phase (:680, :821).  Unmodified, on the
drop-in rasterizer, one such iteration costs ~5 ms at BASELINE config B: ~100 small PyTorch launches around two rasterizer
forwards and two backwards.  ``install(module)`` replaces the two names the loops resolve at call time in the module that holds
them -- ``get_loss`` and ``initialize_optimizer`` -- so that the SAME statements run the fused iteration of ``FusedEngine``:

* ``get_loss`` runs ``splat_iter_loss_backward`` (render, loss, every gradient) and returns a 0-dim loss tensor whose
  ``.backward()`` has nothing left to do, ``variables`` updated as the reference updates them (``seen``, ``max_2D_radius``), and the
  weighted ``losses`` dict;
* ``initialize_optimizer`` returns a ``torch.optim.Adam`` (same ``param_groups`` / ``state`` layout, so the reference's own
  ``remove_points`` / ``cat_params_to_optimizer`` / ``update_params_and_optimizer`` keep working on it: they slice and re-attach
  ``exp_avg`` / ``exp_avg_sq``) whose ``step()`` is one Adam kernel over the engine's gradients, on those very moment tensors.
  A parameter the caller re-created between ``backward()`` and ``step()`` (pruning: a fresh nn.Parameter without ``.grad``) is
  skipped by that step, as torch skips it.

The caller's dict of nn.Parameters stays the single copy of the map: the engine reads and updates it in place, and is rebuilt when
the caller replaces the tensors (pruning, add_new_gaussians).  What the plug-in does NOT provide: ``variables['means2D'].grad``
(gradient-based densification: ``use_gaussian_splatting_densification`` raises), non-zero learning rates for the Gaussians while
tracking or for the poses while mapping (no shipped configuration has them; they raise).  Every ``get_loss`` checks the
iteration's list capacity flags (one 16-byte read) and transparently repeats an iteration whose lists overflowed.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _capi
from .fused import PARAM_ORDER, FusedEngine

_POSE_KEYS = ("cam_unnorm_rots", "cam_trans")


class _FusedLoss(torch.Tensor):
    """The loss value of an iteration whose gradients already exist: ``backward()`` is a no-op."""

    def backward(self, *args, **kwargs):          # noqa: D401
        return None


def _key(params, variables, cam):
    ptrs = tuple(int(params[k].data_ptr()) for k in PARAM_ORDER + _POSE_KEYS)
    shapes = tuple(tuple(params[k].shape) for k in PARAM_ORDER + _POSE_KEYS)
    r = variables.get('max_2D_radius') if variables is not None else None
    return ptrs, shapes, (int(r.data_ptr()) if r is not None else 0), (int(cam.image_height), int(cam.image_width))


class _Session:
    """Engines for the caller's current tensors (one per camera resolution), the phase the next iterations belong to."""

    def __init__(self):
        self.engines = {}           # key -> FusedEngine
        self.current = None         # (key, engine, params dict) of the last get_loss
        self.pending_tracking = True
        self.map_step = 0
        self.stats = {"iterations": 0, "rebuilds": 0, "repeats": 0}

    def engine(self, params, variables, cam):
        key = _key(params, variables, cam)
        eng = self.engines.get(key)
        if eng is None:
            # the caller replaced its tensors (first call, pruning, densification): engines of the old tensors are dead
            self.engines = {k: e for k, e in self.engines.items() if k[0] == key[0] and k[1] == key[1] and k[2] == key[2]}
            eng = FusedEngine(params, cam, track_max_radius=None if variables is None else variables.get('max_2D_radius'))
            self.engines[key] = eng
            self.stats["rebuilds"] += 1
        return key, eng


_session = _Session()


def session_stats():
    return dict(_session.stats)


def get_loss(params, curr_data, variables, iter_time_idx, loss_weights, use_sil_for_loss, sil_thres, use_l1,
             ignore_outlier_depth_loss, tracking=False, mapping=False, do_ba=False, plot_dir=None, visualize_tracking_loss=False,
             tracking_iteration=None):
    """Signature of /root/reference/scripts/splatam.py:214-216; returns (loss, variables, weighted_losses) like the reference."""
    if visualize_tracking_loss:
        raise NotImplementedError("visualize_tracking_loss needs the rendered images of the reference's get_loss; use the drop-in path")
    s = _session
    key, eng = s.engine(params, variables, curr_data['cam'])
    cfg = dict(loss_weights=loss_weights, use_sil_for_loss=use_sil_for_loss, sil_thres=sil_thres, use_l1=use_l1,
               ignore_outlier_depth_loss=ignore_outlier_depth_loss)
    if tracking and (s.pending_tracking or eng.track_time_idx != int(iter_time_idx)):
        eng.begin_tracking(iter_time_idx)           # fresh pose Adam state: the caller made a new optimizer for this frame (:680)
        s.pending_tracking = False
    for attempt in range(3):
        eng.loss_backward(curr_data, iter_time_idx, cfg, tracking=bool(tracking), do_ba=bool(do_ba))
        if not eng.check_overflow():                # a 16-byte read; True: the lists did not fit, they were re-sized / re-learnt
            break
        s.stats["repeats"] += 1
    else:
        raise RuntimeError("the instance lists overflowed three times in a row")
    s.stats["iterations"] += 1
    s.current = (key, eng, params)
    d = eng.buf['d_cam']
    loss = d[7].clone().as_subclass(_FusedLoss)
    depth_w = loss_weights['depth'] * d[8].clone()
    losses = {'depth': depth_w, 'im': loss.as_subclass(torch.Tensor) - depth_w}
    if variables is not None:
        variables['seen'] = eng.seen
    return loss, variables, losses


class FusedOptimizer(torch.optim.Adam):
    """torch.optim.Adam as initialize_optimizer builds it (/root/reference/scripts/splatam.py:160-166), stepping through the engine."""

    def __init__(self, params, lrs_dict, tracking):
        groups = [{'params': [v], 'name': k, 'lr': lrs_dict[k]} for k, v in params.items()]
        if tracking:
            super().__init__(groups)
        else:
            super().__init__(groups, lr=0.0, eps=1e-15)
        self._tracking = bool(tracking)
        self._lrs = dict(lrs_dict)
        if tracking:
            if any(float(lrs_dict[k]) != 0.0 for k in PARAM_ORDER):
                raise NotImplementedError("the fused tracking iteration forms the pose gradient only: Gaussian learning rates must be 0 "
                                          "(as in every shipped configuration)")
            _session.pending_tracking = True
        else:
            if any(float(lrs_dict.get(k, 0.0)) != 0.0 for k in _POSE_KEYS):
                raise NotImplementedError("pose learning rates in the mapping optimizer (bundle adjustment) are not supported by the plug-in")
            _session.map_step = 0
            # the state the reference's map edits expect to find and re-attach (exp_avg / exp_avg_sq per parameter)
            for g in self.param_groups:
                p = g['params'][0]
                if g['name'] in PARAM_ORDER:
                    self.state[p] = {'step': torch.tensor(0.0), 'exp_avg': torch.zeros_like(p), 'exp_avg_sq': torch.zeros_like(p)}

    def zero_grad(self, set_to_none=True):
        return None

    @torch.no_grad()
    def step(self, closure=None):
        s = _session
        if s.current is None:
            raise RuntimeError("optimizer.step() before any get_loss()")
        key, eng, params = s.current
        if self._tracking:
            eng.adam_pose(self._lrs['cam_unnorm_rots'], self._lrs['cam_trans'])
            return None
        by_name = {g['name']: g for g in self.param_groups}
        skip = []
        for idx, k in enumerate(PARAM_ORDER):
            p = by_name[k]['params'][0]
            st = self.state.get(p)
            # a parameter re-created since the backward pass has no gradient: torch's step skips it (pruning, opacity reset).  Compared
            # with the tensors get_loss SAW (the key): the reference's map edits update the caller's dict in place, and the engine
            # reads that same dict
            if int(p.data_ptr()) != key[0][idx] or tuple(p.shape) != key[1][idx] or st is None:
                skip.append(k)
                continue
            eng.exp_avg[k], eng.exp_avg_sq[k] = st['exp_avg'], st['exp_avg_sq']
        if len(skip) == len(PARAM_ORDER):
            return None
        eng.map_step = s.map_step
        o = eng._adam_map_args(self._lrs)
        s.map_step = eng.map_step
        for idx, k in enumerate(PARAM_ORDER):
            if k in skip:
                o.grad[idx] = None
            else:
                self.state[by_name[k]['params'][0]]['step'] += 1
        m = eng._map_struct()
        with torch.cuda.device(eng.dev):
            _capi.check(eng.L.splat_iter_adam_map(C.byref(m), C.byref(o), eng._stream()), "splat_iter_adam_map")
        return None


def initialize_optimizer(params, lrs_dict, tracking):
    return FusedOptimizer(params, lrs_dict, tracking)


class _Installed:
    def __init__(self, module, saved):
        self.module, self.saved = module, saved

    def uninstall(self):
        for k, v in self.saved.items():
            setattr(self.module, k, v)
        _session.engines.clear()
        _session.current = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.uninstall()


def install(module):
    """Replace ``module.get_loss`` and ``module.initialize_optimizer`` (the reference's ``scripts/splatam.py`` module, or any module
    shaped like it, e.g. ``splatam_amd.slam``) by the fused adapters.  Returns a handle with ``uninstall()`` (also a context manager)."""
    saved = {k: getattr(module, k) for k in ("get_loss", "initialize_optimizer") if hasattr(module, k)}
    if len(saved) != 2:
        raise RuntimeError(f"{module!r} does not look like scripts/splatam.py: it has no get_loss / initialize_optimizer")
    module.get_loss = get_loss
    module.initialize_optimizer = initialize_optimizer
    _session.engines.clear()
    _session.current = None
    _session.stats.update(iterations=0, rebuilds=0, repeats=0)
    return _Installed(module, saved)
