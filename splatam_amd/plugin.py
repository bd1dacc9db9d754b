"""Run-time plug-in: the fused iteration under the reference's OWN loop statements.

``scripts/splatam.py``'s tracking and mapping loops (/root/reference/scripts/splatam.py:690-711, :846-869) are

    loss, variables, losses = get_loss(params, data, variables, iter_time_idx, ...)
    loss.backward()
    [prune_gaussians / densify between backward and step, mapping only]
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)

with ``optimizer = initialize_optimizer(params, lrs, tracking=...)`` created per frame and phase (:680, :821), and around them
``add_new_gaussians`` (:378-420) and ``prune_gaussians`` / ``remove_points`` (/root/reference/utils/slam_external.py:139-188)
REPLACE every parameter tensor.  Unmodified, on the drop-in rasterizer, one iteration costs ~5 ms at BASELINE config B: ~100 small
PyTorch launches around two rasterizer forwards and two backwards.  ``install(module)`` replaces the two names the loops resolve at
call time in the module that holds them -- ``get_loss`` and ``initialize_optimizer`` -- so that the SAME statements run the fused
iteration of ``FusedEngine``:

* ``get_loss`` runs ``splat_iter_loss_backward`` (render, loss, every gradient) and returns a 0-dim loss tensor whose
  ``.backward()`` has nothing left to do, ``variables`` updated as the reference updates them (``seen``, ``max_2D_radius``), and the
  weighted ``losses`` dict (``depth``, ``im``, ``loss``: the three values the reference returns, :339-346);
* ``initialize_optimizer`` returns a ``torch.optim.Adam`` (same ``param_groups`` / ``state`` layout, so the reference's own
  ``remove_points`` / ``cat_params_to_optimizer`` / ``update_params_and_optimizer`` keep working on it: they slice and re-attach
  ``exp_avg`` / ``exp_avg_sq``) whose ``step()`` is one Adam kernel over the engine's gradients, on those very moment tensors,
  with torch's per-parameter step counts.  A parameter the caller re-created between ``backward()`` and ``step()`` (pruning: a
  fresh nn.Parameter without ``.grad``) is skipped by that step, as torch skips it.

The caller's dict of nn.Parameters stays the single copy of the map: the engine reads and updates it in place.  When the caller
replaces the tensors the engine is RE-BOUND (``FusedEngine.rebind``): workspace, list statistics and bucket stride survive an edit
that moves the row count by less than 10 %.

No blocking read per iteration.  The per-tile lists have a fixed capacity; an iteration whose lists did not fit raises a flag ON THE
DEVICE and every Adam kernel skips while it is up (include/splat_hip.h, ``SplatIterWorkspace.d_cam[12]``), so a bad iteration never
moves the map or the pose.  The host learns of it
  * from the iteration's 128-byte report, copied to pinned memory asynchronously after every ``get_loss`` and looked at (event
    query, no wait) at the start of later calls: the lists are then re-sized and the loop goes on (mapping: the skipped iterations
    are lost, ``session_stats()['skipped_iterations']``);
  * or the moment the caller reads the loss value anyway (the tracking loop's ``if loss < current_min_loss``, :708): that ONE
    read fetches the report too, and a flagged tracking iteration is repeated on the spot, invisibly to the caller.

Learning rates the shipped configurations leave at zero are honoured too: pose learning rates in the mapping optimizer (bundle
adjustment, ``get_loss(..., do_ba=True)``: the pose gradient of the iteration's report goes through torch's own Adam on the two small
camera tensors) and Gaussian learning rates in the tracking optimizer (rgb / opacity / scale gradients are then formed -- centres and
rotations are detached while tracking, /root/reference/utils/slam_helpers.py:266-271 -- and stepped by the Adam kernel with the
tracking optimizer's eps).  Gradient-based densification (``use_gaussian_splatting_densification``,
/root/reference/scripts/splatam.py:863-866): ``variables['means2D']`` is an object whose ``.grad`` -- the colour pass' screen-space
gradient the reference's ``accumulate_mean2d_gradient`` reads (/root/reference/utils/slam_external.py:100-104) -- is formed on first
access by one extra RGB-only backward composite over the iteration's lists; configurations that never read it never pay for it.
What the plug-in does NOT provide: ``visualize_tracking_loss``.

``install(module, map_edits=True)`` (opt-in) replaces ``add_new_gaussians`` and ``prune_gaussians`` too: ONE engine then owns a
capacity-managed map for the run and edits it in place (``splat_map_add_new_gaussians`` / ``splat_map_prune``), the caller's dict
entries are views of its rows, re-pointed where the reference replaces them, and the optimizer object carries no per-tensor state
(see ``install``).  The loop statements are the reference's in both modes.
"""
from __future__ import annotations

import ctypes as C
import operator

import torch

from . import _capi
from .fused import PARAM_ORDER, FusedEngine

_POSE_KEYS = ("cam_unnorm_rots", "cam_trans")
_ALL_KEYS = PARAM_ORDER + _POSE_KEYS


class _Report:
    """One iteration's report (``d_cam``, SPLAT_ITER_DCAM floats) on the device, and its host copy once someone needed it."""
    __slots__ = ("dev", "host", "host_tensor", "args", "stepped", "digested", "optimizer")

    def __init__(self, dev):
        self.dev = dev              # device copy (the engine's own buffer is overwritten by the next iteration)
        self.host = None            # list of floats, after the first read
        self.host_tensor = None     # the same 128 bytes as a CPU tensor (its int32 fields are read as int32: _report_values)
        self.optimizer = None       # the mapping optimizer that stepped (or tried to) on this iteration
        self.args = None            # what get_loss was called with (a flagged tracking iteration is repeated from it)
        self.stepped = False        # optimizer.step() ran on this iteration
        self.digested = False


class _FusedLoss(torch.Tensor):
    """The loss value of an iteration whose gradients already exist: ``backward()`` is a no-op.  Reading the VALUE (comparison,
    float(), item(), format) costs one device read, which also brings the iteration's capacity flags: see the module docstring."""

    # torch functions see (and return) plain tensors: the overrides below are the whole of the subclass
    __torch_function__ = torch._C._disabled_torch_function_impl

    def backward(self, *args, **kwargs):          # noqa: D401
        return None

    # -- value reads: through the report, so that the flag check rides on the read the caller does anyway
    def _value(self):
        rep = getattr(self, "_report", None)
        if rep is None:
            return float(self.as_subclass(torch.Tensor))
        return _session.value_of(rep, getattr(self, "_slot", 7))

    def item(self):
        return self._value()

    def __float__(self):
        return self._value()

    def __format__(self, spec):
        return format(self._value(), spec)

    # comparisons with numbers and with other losses are decided on the host (the reference's `if loss < current_min_loss` reads the
    # value there anyway); with any other tensor they stay torch operations on the device
    def _cmp(self, o, op, name):
        if isinstance(o, _FusedLoss):
            return op(self._value(), o._value())
        if torch.is_tensor(o):
            return getattr(torch.Tensor, name)(self.as_subclass(torch.Tensor), o)
        return op(self._value(), o)

    def __lt__(self, o):
        return self._cmp(o, operator.lt, "__lt__")

    def __le__(self, o):
        return self._cmp(o, operator.le, "__le__")

    def __gt__(self, o):
        return self._cmp(o, operator.gt, "__gt__")

    def __ge__(self, o):
        return self._cmp(o, operator.ge, "__ge__")


def _report_values(host):
    """The report as Python numbers: float slots as floats, the int32 slots [16..21] (status words, flagged bit, skipped count) as
    ints read THROUGH AN int32 VIEW -- their bit patterns are denormal as floats and would flush to zero under FTZ / DAZ
    (torch.set_flush_denormal, a library built with -ffast-math) if they travelled as floats."""
    vals = host.tolist()
    ints = host.view(torch.int32)[16:22].tolist()
    vals[16:22] = ints
    return vals


class _Means2D:
    """``variables['means2D']`` of a plug-in iteration: the reference keeps the (zero) render variable there to read its ``.grad``
    after ``backward()`` (/root/reference/scripts/splatam.py:248-250, utils/slam_external.py:100-104).  Here ``.grad`` is the colour
    pass' dL/dmeans2D [P, 3] (third column zero, as the rasterizer returns it), formed on first access from the iteration's
    workspace -- valid until the next ``get_loss`` of the same engine."""

    def __init__(self, eng, rep, tracking):
        self._eng, self._rep, self._tracking, self._grad = eng, rep, tracking, None
        self.shape = (eng.P, 3)

    def retain_grad(self):
        return None

    @property
    def grad(self):
        if self._grad is None:
            cur = _session.current
            if cur is None or cur[2] is not self._rep:
                raise RuntimeError("variables['means2D'].grad is formed from the iteration's workspace: read it before the next get_loss()")
            if self._tracking:
                raise RuntimeError("variables['means2D'].grad of a TRACKING iteration: the fused tracking composite keeps no planes "
                                   "(the reference reads this gradient in mapping only: densify, scripts/splatam.py:863-866)")
            g2 = self._eng.means2d_gradient()
            self._grad = torch.cat((g2, torch.zeros(g2.shape[0], 1, device=g2.device)), dim=1)
        return self._grad


def _scalar(rep, slot):
    t = rep.dev[slot].as_subclass(_FusedLoss)
    t._report, t._slot = rep, slot
    return t


class _Session:
    """Engines for the caller's current tensors (one per camera resolution), the phase the next iterations belong to."""

    def __init__(self):
        self.engines = {}           # (H, W) -> FusedEngine
        self.bound = {}             # (H, W) -> tuple of the tensor OBJECTS the engine is bound to (kept alive: ids stay unique)
        self.current = None         # (engine, bound tensors, report) of the last get_loss
        self.pending_tracking = True
        self.track_steps_gaussians = False      # the live tracking optimizer has non-zero Gaussian learning rates
        self.pending = []           # [(engine, report, pinned host tensor, event)] reports in flight to the host
        self.pool = []              # pinned buffers / events for re-use
        self.stats = {"iterations": 0, "rebuilds": 0, "engines_built": 0, "repeats": 0, "skipped_iterations": 0}
        self.map_edits = False      # install(module, map_edits=True): the engine OWNS the map (capacity-managed, edited in place)
        self.skip_gaussian_step = False     # map_edits: an edit re-created the parameters since backward(): the next step() leaves them alone
        self.pending_map_reset = False      # map_edits: a mapping optimizer was created before the engine existed

    # ---------------------------------------------------------------- engines
    def managed_engine(self, params, variables, cam):
        """map_edits mode: ONE engine that owns the Gaussian tensors (FusedEngine(gaussian_capacity=...)): the caller's dict entries
        are views of the first P rows of its backing arrays, re-pointed by the adapters of add_new_gaussians / prune_gaussians exactly
        where the reference replaces them.  A tensor replaced behind the engine's back (an un-adapted edit) is an error, not a re-bind."""
        hw = (int(cam.image_height), int(cam.image_width))
        eng = self.engines.get(hw)
        if eng is None:
            if self.engines:
                raise NotImplementedError("plugin.install(map_edits=True): one camera resolution per run (the engine owns the map)")
            if variables is None:
                raise RuntimeError("plugin.install(map_edits=True) needs the caller's `variables` dict (the per-Gaussian variables move with the rows)")
            P = int(params['means3D'].shape[0])
            cap = max(int(1.5 * P), int(2.5 * hw[0] * hw[1])) + 65536
            eng = FusedEngine(params, cam, gaussian_capacity=cap, variables=variables)
            self.engines[hw] = eng
            self.stats["engines_built"] += 1
            self.stats["rebuilds"] += 1
            if self.pending_map_reset:
                eng.reset_map_optimizer()
                self.pending_map_reset = False
        elif eng.params is not params or any(params[k].data_ptr() != eng.store[k].data_ptr() or params[k].shape[0] != eng.P for k in PARAM_ORDER):
            raise RuntimeError("plugin.install(map_edits=True): the Gaussian tensors were replaced outside add_new_gaussians / prune_gaussians; "
                               "the engine owns the map in this mode")
        now = tuple(params[k] for k in _ALL_KEYS) + (variables.get('max_2D_radius') if variables is not None else None,)
        self.bound[hw] = now
        return eng, now

    def engine(self, params, variables, cam):
        if self.map_edits:
            return self.managed_engine(params, variables, cam)
        hw = (int(cam.image_height), int(cam.image_width))
        radius = variables.get('max_2D_radius') if variables is not None else None
        now = tuple(params[k] for k in _ALL_KEYS) + (radius,)
        eng = self.engines.get(hw)
        was = self.bound.get(hw)
        if eng is not None and was is not None and all(a is b for a, b in zip(now, was)):
            return eng, was
        # the caller replaced its tensors (first call, pruning, add_new_gaussians): re-bind, or build the first engine of this size
        if eng is not None and ((params['log_scales'].shape[1] == 1) != eng.iso or params['cam_unnorm_rots'].shape[-1] != eng.num_frames
                                or params['means3D'].device != eng.dev):
            eng = None
        if eng is None:
            eng = FusedEngine(params, cam, track_max_radius=radius, row_headroom=0.125)
            self.engines[hw] = eng
            self.stats["engines_built"] += 1
        else:
            self.drain(eng)                     # reports of the old tensors' iterations: learn from them before the lists are judged
            eng.rebind(params, radius)
        self.bound[hw] = now
        self.stats["rebuilds"] += 1
        return eng, now

    # ---------------------------------------------------------------- reports
    def post(self, eng, rep):
        """Queue the asynchronous host copy of a report."""
        if self.pool:
            host, ev = self.pool.pop()
        else:
            host, ev = torch.empty(_capi.SPLAT_ITER_DCAM, dtype=torch.float32, pin_memory=True), torch.cuda.Event()
        host.copy_(rep.dev, non_blocking=True)
        ev.record()
        self.pending.append((eng, rep, host, ev))

    def poll(self, wait=False):
        """Look at the reports that have arrived (``wait``: at all of them)."""
        while self.pending:
            eng, rep, host, ev = self.pending[0]
            if not wait and not ev.query():
                return
            if wait:
                ev.synchronize()
            self.pending.pop(0)
            if rep.host is None:
                rep.host = _report_values(host)
            self.digest(eng, rep, host)
            self.pool.append((host, ev))

    def drain(self, eng=None):
        self.poll(wait=True)

    def digest(self, eng, rep, host=None):
        """Learn the list statistics / deal with a capacity flag from one report (once)."""
        if rep.digested:
            return False
        rep.digested = True
        if host is None:
            host = rep.host_tensor
        # a flagged report: the iterations of this engine that were launched behind it (their reports are still in flight) were gated on
        # the device too -- every one of them raised the skipped count d_cam[21], and the host-side step counts advanced with every
        # step().  The NEWEST of those reports carries the total: wait for it BEFORE the flag is cleared (digest_report queues the
        # clear behind them), so that all of them are taken back, not just the first
        flagged = float(host[12]) != 0.0 or any(int(v) != 0 for v in host.view(torch.int32)[[17, 19]].tolist())
        skipped_in_flight = 0
        if flagged:
            for e2, r2, h2, ev2 in reversed(self.pending):
                if e2 is eng and r2 is not rep:
                    ev2.synchronize()
                    skipped_in_flight = int(h2.view(torch.int32)[21])
                    break
        if eng.digest_report(host):
            skipped = max(eng.skipped_iterations, skipped_in_flight)
            eng.skipped_iterations = skipped
            self.stats["skipped_iterations"] += skipped
            self.lost_steps(rep, skipped)
            # reports already in flight were written under the same flag: they carry nothing new
            for _, r, _, _ in self.pending:
                r.digested = True
            return True
        return False

    def lost_steps(self, rep, skipped):
        """A flagged MAPPING iteration (and every one until the host learnt of it) took no Adam step on the device, but the host-side
        step counts of the optimizer advanced with every ``step()``: take them back, so that the bias corrections of the following
        steps match the moments on the device.  The lost iterations are not repeated (the caller's loop has moved on: another
        keyframe, another iteration index); they are counted in ``session_stats()['skipped_iterations']`` and warned about once."""
        opt = rep.optimizer
        if opt is None or rep.args is None or rep.args[-1] or skipped <= 0:
            return                                              # (tracking: repeat_tracking re-runs the iteration instead)
        if self.map_edits:
            for eng in self.engines.values():
                eng.map_step = max(eng.map_step - int(skipped), 0)
        for g in opt.param_groups:             # (the pose groups carry state only when bundle adjustment steps them)
            st = opt.state.get(g['params'][0])
            if st is not None and 'step' in st:
                st['step'] = torch.clamp(st['step'] - float(skipped), min=0.0)
        if not self.stats.get("warned_lost_steps"):
            self.stats["warned_lost_steps"] = True
            import warnings
            warnings.warn(f"splatam_amd.plugin: {skipped} mapping iteration(s) ran on per-tile lists that did not fit and took no Adam "
                          "step (the lists have been re-sized); see session_stats()['skipped_iterations']")

    def value_of(self, rep, slot):
        """The caller reads a value of an iteration's report: ONE device read fetches the whole report, flags included."""
        if rep.host is None:
            # ONE wait: the whole report, flags included.  Its asynchronous copy was queued right behind the iteration's kernels
            # (post()): wait for THAT (the earliest moment the values exist on the host) instead of queueing a second, blocking copy
            # behind whatever the caller has launched since (the pose's Adam step)
            for _, r, host, ev in self.pending:
                if r is rep:
                    ev.synchronize()
                    rep.host_tensor = host.clone()              # (the pinned buffer goes back to the pool when poll() reaches it)
                    break
            else:
                rep.host_tensor = rep.dev.cpu()
            rep.host = _report_values(rep.host_tensor)
            eng = self.current[0] if self.current is not None and self.current[2] is rep else None
            if eng is not None and self.digest(eng, rep) and rep.args is not None and rep.args[-1]:
                self.repeat_tracking(eng, rep)
        return rep.host[slot]

    def repeat_tracking(self, eng, rep):
        """The tracking iteration the caller is looking at ran on lists that did not fit: its Adam step was skipped on the device.
        The lists have been re-sized (digest); run it again -- same pose, same frame -- and, when the caller has already stepped,
        take that step now.  The caller sees the repeated iteration's values."""
        params, curr_data, iter_time_idx, cfg, do_ba, tracking = rep.args
        for _ in range(3):
            eng.loss_backward(curr_data, iter_time_idx, cfg, tracking=True, do_ba=do_ba, keep_planes=False, map_grads=bool(self.track_steps_gaussians))
            if rep.stepped is not False:
                lr, opt, bound = rep.stepped
                eng.pose_step -= 1
                eng.adam_pose(*lr)
                if opt is not None:                 # (the skipped step counted: take it back, then step)
                    for g in opt.param_groups:
                        st = opt.state.get(g['params'][0]) if g['name'] in ("rgb_colors", "logit_opacities", "log_scales") else None
                        if st is not None:
                            st['step'] -= 1
                    opt._step_gaussians(eng, bound, eps=1e-8, tracking=True)
            fresh_t = eng.buf['d_cam'].cpu()
            fresh = _report_values(fresh_t)
            self.stats["repeats"] += 1
            if fresh[12] == 0.0:
                rep.dev.copy_(eng.buf['d_cam'])
                rep.host, rep.host_tensor = fresh, fresh_t
                return
            eng.digest_report(fresh_t)
        raise RuntimeError("the instance lists overflowed three times in a row")


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
    s.poll()                                        # reports that have arrived since the last call (no wait)
    eng, bound = s.engine(params, variables, curr_data['cam'])
    cfg = dict(loss_weights=loss_weights, use_sil_for_loss=use_sil_for_loss, sil_thres=sil_thres, use_l1=use_l1,
               ignore_outlier_depth_loss=ignore_outlier_depth_loss)
    tracking = bool(tracking)
    if tracking and (s.pending_tracking or eng.track_time_idx != int(iter_time_idx)):
        eng.begin_tracking(iter_time_idx)           # fresh pose Adam state: the caller made a new optimizer for this frame (:680)
        s.pending_tracking = False
    # (a tracking optimizer with non-zero Gaussian learning rates: the gradients torch would step -- rgb, opacity, scale -- are wanted)
    want_map = True if not tracking else bool(s.track_steps_gaussians)
    eng.loss_backward(curr_data, iter_time_idx, cfg, tracking=tracking, do_ba=bool(do_ba), keep_planes=False, map_grads=want_map)
    rep = _Report(eng.buf['d_cam'].clone())
    rep.args = (params, curr_data, int(iter_time_idx), cfg, bool(do_ba), tracking)
    s.post(eng, rep)
    s.stats["iterations"] += 1
    s.current = (eng, bound, rep)
    loss = _scalar(rep, 7)
    losses = {'im': _scalar(rep, 15), 'loss': loss}
    if use_l1:
        losses['depth'] = _scalar(rep, 14)
    if variables is not None:
        variables['seen'] = eng.seen
        variables['means2D'] = _Means2D(eng, rep, tracking)
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
        s = _session
        # a phase boundary: the reference's own statements have just synchronised (add_new_gaussians' boolean indexing, the loss
        # comparison of the last tracking iteration); every report of the finished phase is looked at here
        s.drain()
        # groups of the OTHER kind with a non-zero learning rate (no shipped configuration): Gaussians while tracking, poses while mapping
        self._steps_gaussians = (not tracking) or any(float(lrs_dict.get(k, 0.0)) != 0.0 for k in PARAM_ORDER)
        self._steps_poses = tracking or any(float(lrs_dict.get(k, 0.0)) != 0.0 for k in _POSE_KEYS)
        if tracking:
            s.pending_tracking = True
            s.track_steps_gaussians = self._steps_gaussians
        self._managed = bool(s.map_edits)
        if self._managed:
            # the engine owns parameters and moments: nothing is keyed by tensor here (the reference's own map edits, which slice
            # optimizer.state, are replaced by the adapters below)
            if tracking and self._steps_gaussians:
                raise NotImplementedError("plugin.install(map_edits=True): Gaussian learning rates in the TRACKING optimizer are not supported")
            if not tracking:
                if s.engines:
                    for eng in s.engines.values():
                        eng.reset_map_optimizer()           # a new optimizer: fresh moments, step count 0 (:821)
                else:
                    s.pending_map_reset = True
                s.skip_gaussian_step = False
            return
        if self._steps_gaussians:
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
        eng, bound, rep = s.current
        if self._tracking:
            lr = (self._lrs['cam_unnorm_rots'], self._lrs['cam_trans'])
            eng.adam_pose(*lr)
            rep.stepped = (lr, self if self._steps_gaussians else None, bound)
            if self._steps_gaussians:
                self._step_gaussians(eng, bound, eps=1e-8, tracking=True)       # torch.optim.Adam(param_groups): default eps
            return None
        rep.optimizer = self
        if self._managed:
            # an edit between backward() and step() re-created the parameters (remove_points does, whether or not a row goes):
            # torch finds no .grad on them and leaves them alone -- so does this step
            if s.skip_gaussian_step:
                s.skip_gaussian_step = False
            else:
                eng.adam_map(self._lrs)
            if self._steps_poses and rep.args is not None and rep.args[4]:
                self._step_poses_with_torch(rep)
            rep.stepped = True
            return None
        self._step_gaussians(eng, bound, eps=1e-15, tracking=False)
        if self._steps_poses and rep.args is not None and rep.args[4]:
            self._step_poses_with_torch(rep)
        rep.stepped = True
        return None

    def _step_gaussians(self, eng, bound, eps, tracking):
        by_name = {g['name']: g for g in self.param_groups}
        steps, live = [1] * 5, []
        for idx, k in enumerate(PARAM_ORDER):
            p = by_name[k]['params'][0]
            st = self.state.get(p)
            # a parameter re-created since the backward pass has no gradient: torch's step skips it (pruning, opacity reset).  Compared
            # with the tensors get_loss SAW: the reference's map edits update the caller's dict in place, and the engine reads that dict.
            # While tracking the centres and rotations are detached (no gradient: skipped as torch skips them)
            if p is not bound[idx] or st is None or (tracking and k in ("means3D", "unnorm_rotations")):
                continue
            eng.exp_avg[k], eng.exp_avg_sq[k] = st['exp_avg'], st['exp_avg_sq']
            st['step'] += 1                         # torch counts per parameter (a CPU scalar tensor, as torch keeps it)
            steps[idx] = int(st['step'])
            live.append(idx)
        if not live:
            return
        o = eng._adam_map_args(self._lrs, eps=eps, steps=steps)
        for idx in range(5):
            if idx not in live:
                o.grad[idx] = None
        m = eng._map_struct()
        with torch.cuda.device(eng.dev):
            _capi.check(eng.L.splat_iter_adam_map(C.byref(m), C.byref(o), eng._stream()), "splat_iter_adam_map")

    def _step_poses_with_torch(self, rep):
        """Bundle adjustment (pose learning rates in the mapping optimizer, get_loss(do_ba=True)): the iteration's pose gradient -- one
        column of the two camera tensors -- through torch's own Adam step for those two parameters (every column moves with its
        moments, as torch moves it).  A FLAGGED iteration takes no step at all -- poses, moments and step counts stay as they were
        (decided on the device: the values after the step are kept only where the report's flag is down; no host read)."""
        by_name = {g['name']: g for g in self.param_groups}
        t = rep.args[2]
        ok = rep.dev[12] == 0
        pose, before = [], []
        for name, lo, hi in (('cam_unnorm_rots', 0, 4), ('cam_trans', 4, 7)):
            p = by_name[name]['params'][0]
            g = torch.zeros_like(p)
            g[0, :, t] = rep.dev[lo:hi]
            p.grad = g
            pose.append(p)
            st = self.state.get(p, {})
            before.append((p.detach().clone(), {k: v.clone() for k, v in st.items() if torch.is_tensor(v)}))
        saved = [(g, g['params']) for g in self.param_groups]
        try:
            for g in self.param_groups:             # torch steps the parameters that carry a .grad: only the two camera tensors here
                if g['name'] not in _POSE_KEYS:
                    g['params'] = []
            torch.optim.Adam.step(self)
        finally:
            for g, params in saved:
                g['params'] = params
            for p, (old, old_state) in zip(pose, before):
                p.grad = None
                p.data.copy_(torch.where(ok, p.data, old))
                for k, v in self.state.get(p, {}).items():
                    if torch.is_tensor(v) and v.device.type != "cpu":       # exp_avg, exp_avg_sq (torch keeps `step` on the host: lost_steps)
                        prev = old_state.get(k)
                        v.copy_(torch.where(ok, v, torch.zeros_like(v) if prev is None else prev))


def initialize_optimizer(params, lrs_dict, tracking):
    return FusedOptimizer(params, lrs_dict, tracking)


def _refresh_bound(s, eng, params, variables):
    hw = (eng.H, eng.W)
    now = tuple(params[k] for k in _ALL_KEYS) + (variables.get('max_2D_radius') if variables is not None else None,)
    s.bound[hw] = now
    if s.current is not None and s.current[0] is eng:
        s.current = (eng, now, s.current[2])


def add_new_gaussians(params, variables, curr_data, sil_thres, time_idx, mean_sq_dist_method, gaussian_distribution):
    """map_edits mode: /root/reference/scripts/splatam.py:378-420 as an in-place edit of the engine's map (render at the tracked pose,
    non-presence masks with the exact median, one Gaussian per selected pixel appended in pixel order, variables reset as the reference
    resets them); ``params`` / ``variables`` are the caller's dicts, their entries re-pointed at the grown arrays."""
    s = _session
    s.drain()
    eng, _ = s.engine(params, variables, curr_data['cam'])
    eng.add_new_gaussians(curr_data, sil_thres, time_idx, mean_sq_dist_method, gaussian_distribution)
    if not eng.lists_known():
        eng.relearn_lists(curr_data, time_idx)          # the per-tile lists of the grown map (one probe render + one read)
    _refresh_bound(s, eng, params, variables)
    s.stats["rebuilds"] += 1
    return params, variables


def prune_gaussians(params, variables, optimizer, iter, prune_dict):
    """map_edits mode: /root/reference/utils/slam_external.py:169-196 (+ remove_points :139-162) as an in-place stable compaction of the
    engine's map, moments and per-Gaussian variables.  On the pruning schedule the reference re-creates every parameter whether or not
    a row goes, so the ``optimizer.step()`` that follows moves none of them: remembered for that step."""
    s = _session
    on_schedule = iter <= prune_dict['stop_after'] and iter >= prune_dict['start_after'] and iter % prune_dict['prune_every'] == 0
    resets = iter <= prune_dict['stop_after'] and iter > 0 and iter % prune_dict['reset_opacities_every'] == 0 and prune_dict['reset_opacities']
    if not (on_schedule or resets):
        return params, variables
    if s.current is None:
        raise RuntimeError("prune_gaussians() before any get_loss()")
    eng, _, rep = s.current
    if eng.params is not params:
        raise RuntimeError("prune_gaussians(): not the params dict the engine owns")
    s.drain()
    removed = eng.prune_gaussians(iter, prune_dict, variables['scene_radius'])
    if on_schedule:
        s.skip_gaussian_step = True
    if removed:
        if not eng.lists_known():
            eng.relearn_lists(rep.args[1], rep.args[2])
        s.stats["rebuilds"] += 1
    _refresh_bound(s, eng, params, variables)
    return params, variables


def densify(params, variables, optimizer, iter, densify_dict):
    raise NotImplementedError("plugin.install(map_edits=True) does not adapt densify(): run gradient-based densification with map_edits=False "
                              "(variables['means2D'].grad is served) or on pipeline.rgbd_slam(engine='fused')")


class _Installed:
    def __init__(self, module, saved):
        self.module, self.saved = module, saved

    def uninstall(self):
        for k, v in self.saved.items():
            setattr(self.module, k, v)
        _reset_session()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.uninstall()


def _reset_session():
    s = _session
    if s.pending:
        try:
            s.poll(wait=True)
        except Exception:                           # noqa: BLE001 (tear-down: the engines are going away)
            s.pending.clear()
    s.engines.clear()
    s.bound.clear()
    s.current = None
    s.map_edits = False
    s.skip_gaussian_step = False
    s.pending_map_reset = False


def install(module, map_edits=False):
    """Replace ``module.get_loss`` and ``module.initialize_optimizer`` (the reference's ``scripts/splatam.py`` module, or any module
    shaped like it, e.g. ``splatam_amd.slam``) by the fused adapters.  Returns a handle with ``uninstall()`` (also a context manager).

    ``map_edits=True`` also replaces ``add_new_gaussians`` and ``prune_gaussians`` (and refuses ``densify``): the engine then OWNS the
    map -- a capacity-managed struct of arrays edited in place on the device, the caller's dict entries re-pointed where the reference
    replaces them -- instead of being re-bound to tensors that torch.cat / boolean indexing re-create three times per frame (at 816 k
    Gaussians: add_new_gaussians 3-10 ms and pruning 3.3 ms per frame against 0.6 + 0.5 ms).  The loop statements stay the reference's.
    One camera resolution per run in this mode (separate tracking / densification resolutions raise)."""
    names = ("get_loss", "initialize_optimizer") + (("add_new_gaussians", "prune_gaussians") if map_edits else ())
    saved = {k: getattr(module, k) for k in names if hasattr(module, k)}
    if len(saved) != len(names):
        raise RuntimeError(f"{module!r} does not look like scripts/splatam.py: it has no {' / '.join(n for n in names if n not in saved)}")
    if map_edits and hasattr(module, "densify"):
        saved["densify"] = module.densify
        module.densify = densify
    module.get_loss = get_loss
    module.initialize_optimizer = initialize_optimizer
    if map_edits:
        module.add_new_gaussians = add_new_gaussians
        module.prune_gaussians = prune_gaussians
    _reset_session()
    _session.map_edits = bool(map_edits)
    _session.stats.update(iterations=0, rebuilds=0, engines_built=0, repeats=0, skipped_iterations=0)
    return _Installed(module, saved)
