"""Autograd front-ends and grouping modules over ``pointnet2._ext``.

API contract of third_party_pointnet2/pointnet2/pointnet2_utils.py:48-420 -- the ``Function`` classes
``FurthestPointSampling``, ``GatherOperation``, ``ThreeNN``, ``ThreeInterpolate``, ``GroupingOperation``,
``BallQuery`` with their lower-case ``.apply`` aliases, and the modules ``QueryAndGroup`` / ``GroupAll`` with
the reference's constructor keywords, argument order and differentiability.  The operators themselves run in
``libcoda_hip.so`` (``_ext.py``); this file only connects them to autograd.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _ext


class _IndexProducer(Function):
    """Base of the operators that only produce indices: nothing is differentiable, so the backward returns
    one ``None`` per forward argument."""

    @classmethod
    def _nones(cls, n):
        return (None,) * n


class FurthestPointSampling(_IndexProducer):
    """(xyz (B,N,3), npoint) -> sampled indices (B,npoint) int32."""

    @staticmethod
    def forward(ctx, xyz, npoint):
        picked = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(picked)
        return picked

    @staticmethod
    def backward(ctx, *_):
        return FurthestPointSampling._nones(2)


furthest_point_sample = FurthestPointSampling.apply


class SamplingPrefetcher:
    """Sampling / grouping of an UPCOMING batch on a side HIP stream.

    FPS is ``npoint`` strictly dependent rounds on one workgroup per scene: for a batch of 8
    scenes it keeps 8 of the 256 CUs busy for milliseconds, and everything else in the model
    waits for its indices.  A training loop knows its next batch while the current step runs,
    so ``submit(point_clouds, module)`` runs the parameter-free front of the set-abstraction
    module (``PointnetSAModuleVotes.prepare``: FPS, centre gather, ball query + grouping,
    distinct-row counts) for that batch on a side stream -- the other 248 CUs keep working on
    the current step -- and the model's forward picks the result up with ``take``: same
    kernels, same values, no host synchronisation on the compute stream.  Because the
    distinct-row count has reached the host by then, the shared MLP can run on de-duplicated
    groups without stalling.  Entries are matched by tensor identity + version, so a batch
    that was modified or never submitted is simply processed in line.

    ``wait_for``: what the side stream has to wait for before it reads the batch -- "current"
    (default: everything enqueued on the caller's stream so far), a ``torch.cuda.Event`` (e.g.
    the batch's host-to-device copy), or None when the batch is known to be resident.

    ``wait_for_counts``: what ``take`` does when the side stream has not finished yet.  False
    (default, right for an eagerly enqueued step whose host thread is the bottleneck): never
    stall the host -- the row counts are dropped and the shared MLP runs on all rows.  True
    (right when the rest of the step is a hipGraph replay and the host has time to spare):
    wait for the side stream's event, at most one sampling time.
    """

    def __init__(self, max_pending=4, wait_for_counts=False):
        self._stream = None
        self._pending = []
        self._max = max_pending
        self.wait_for_counts = wait_for_counts
        self._unchecked = []  # completion events of fronts handed out before the side stream had finished

    def _check_finished(self):
        """A front taken while the side stream was still running was handed to the step UNCHECKED (the host must not
        stall for it).  Its sampling status is read here, at the next submit / take -- one step later at most, and with
        the event known to be complete: a launch that lost its partner workgroup (``CODA_ELOST``) raises now, so the
        step trained on wrong indices is attributed to the batch before this one and can be discarded by the caller."""
        if self._unchecked and all(ev.query() for ev in self._unchecked):
            self._unchecked.clear()
            _ext.check_sampling_status()

    def submit(self, point_clouds, module, wait_for="current", after=None):
        """``after(prepared)``: more work for the side stream that only needs the prepared front (it may add
        entries to the dict), run before the completion event is recorded."""
        dev = point_clouds.device
        self._check_finished()
        if self._stream is None or self._stream.device != dev:
            # A HIGH-PRIORITY stream: it gets a hardware queue of its own.  Streams of equal priority are dealt onto a
            # few hardware queues round-robin, and once a communication library has created its streams the sampling
            # stream can end up behind the SAME queue as the compute stream: the 3.4 ms FPS kernel then serialises
            # with the step instead of running next to it (measured: 350 instead of 444 scenes/s as soon as an RCCL
            # process group existed).  CODA_PREFETCH_PRIORITY=0 restores the default priority (A/B).
            import os
            self._stream = torch.cuda.Stream(device=dev, priority=int(os.environ.get("CODA_PREFETCH_PRIORITY", "-1")))
        if isinstance(wait_for, torch.cuda.Event):
            self._stream.wait_event(wait_for)
        elif wait_for == "current":
            self._stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self._stream), torch.no_grad():
            xyz = point_clouds[..., 0:3].contiguous()
            prepared = module.prepare(xyz)
            if prepared is None:
                return False
            if after is not None:
                after(prepared)
            done = torch.cuda.Event()
            done.record(self._stream)
        self._pending.append((point_clouds, point_clouds._version, prepared, done))
        del self._pending[:-self._max]
        # the sampling kernels hold whole CUs (one 1024-thread workgroup per scene and 20 480 points) for milliseconds:
        # launches that size their grid by the CU count can leave those out (sampling_busy_cus)
        global _ACTIVE_SAMPLING
        _ACTIVE_SAMPLING = (done, min(64, xyz.shape[0] * -(-xyz.shape[1] // 20480)))
        return True

    def take(self, point_clouds):
        """The prepared front of this very tensor (see ``PointnetSAModuleVotes.prepare``), or None."""
        self._check_finished()
        for k, (pc, version, prepared, done) in enumerate(self._pending):
            if pc is point_clouds and version == point_clouds._version:
                del self._pending[k]
                finished = False
                if self.wait_for_counts:
                    done.synchronize()
                    finished = True
                elif not done.query():
                    # the side stream is still busy (its workgroups need whole CUs and may have been
                    # starved): do not stall the host for the row count -- drop it (the shared MLP then
                    # runs on all rows) and let the compute stream wait on the device side
                    prepared = dict(prepared, total_host=None)
                    self._unchecked.append(done)  # status read once the event has completed: _check_finished()
                else:
                    finished = True
                if finished:
                    _ext.check_sampling_status()  # a sampling kernel that lost its partner workgroup is fatal
                cur = torch.cuda.current_stream(point_clouds.device)
                cur.wait_event(done)
                for v in prepared.values():
                    for t in (v if isinstance(v, tuple) else (v,)):
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(cur)
                return prepared
        return None


_ACTIVE_SAMPLING = None  # (completion event, CUs held) of the side stream's latest sampling front


def sampling_busy_cus():
    """CUs a still-running side-stream sampling front holds (0: none in flight).  A persistent kernel launched with
    two workgroups per CU of the WHOLE chip next to it runs the workgroups that found no CU in a second round
    (set-abstraction MLP forward: 425 instead of 312 us); one sized for the free CUs does not."""
    global _ACTIVE_SAMPLING
    if _ACTIVE_SAMPLING is None:
        return 0
    done, cus = _ACTIVE_SAMPLING
    if done.query():
        _ACTIVE_SAMPLING = None
        return 0
    return cus


class GatherOperation(Function):
    """(features (B,C,N), idx (B,npoint)) -> features[..., idx] (B,C,npoint); adjoint = scatter-add."""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n_source = features.shape[2]
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.gather_points_grad(grad_out.contiguous(), idx, ctx.n_source), None


gather_operation = GatherOperation.apply


class ThreeNN(_IndexProducer):
    """(unknown (B,n,3), known (B,m,3)) -> (L2 distances (B,n,3), indices (B,n,3)) of the three nearest
    known points.  The operator returns squared distances; the root is taken here, as in the reference."""

    @staticmethod
    def forward(ctx, unknown, known):
        squared, nearest = _ext.three_nn(unknown, known)
        ctx.mark_non_differentiable(nearest)
        return squared.sqrt(), nearest

    @staticmethod
    def backward(ctx, *_):
        return ThreeNN._nones(2)


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """(features (B,c,m), idx (B,n,3), weight (B,n,3)) -> weighted sum of three features per point (B,c,n)."""

    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.save_for_backward(idx, weight)
        ctx.m_source = features.shape[2]
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m_source), None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """(features (B,C,N), idx (B,npoint,nsample)) -> (B,C,npoint,nsample); adjoint = scatter-add."""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n_source = features.shape[2]
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.group_points_grad(grad_out.contiguous(), idx, ctx.n_source), None


grouping_operation = GroupingOperation.apply


class BallQuery(_IndexProducer):
    """(radius, nsample, xyz (B,N,3), new_xyz (B,npoint,3)) -> neighbour indices (B,npoint,nsample) int32."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        neighbours = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(neighbours)
        return neighbours

    @staticmethod
    def backward(ctx, *_):
        return BallQuery._nones(4)


ball_query = BallQuery.apply


def _pack(primary, *optional):
    """(value, (flag, extra), ...) -> value alone, or a tuple with the requested extras appended."""
    extras = [extra for wanted, extra in optional if wanted]
    return primary if not extras else (primary, *extras)


class QueryAndGroup(nn.Module):
    """Ball query around ``new_xyz`` + grouping of the neighbours' coordinates (relative to their centre,
    optionally in units of the radius) and features.

    When the coordinates carry no gradient (always the case for an input cloud) the coordinate branch is ONE
    fused kernel that emits the indices and the centred / normalised ``grouped_xyz`` together; otherwise the
    separate operators are chained so that autograd sees every step.
    """

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        if ret_unique_cnt and not sample_uniformly:
            raise AssertionError("ret_unique_cnt needs sample_uniformly")
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz, self.ret_unique_cnt = ret_grouped_xyz, ret_unique_cnt
        self.normalize_xyz, self.sample_uniformly = normalize_xyz, sample_uniformly

    def _resample_uniformly(self, idx):
        """Replace the padding of every ball (copies of its first hit) by a uniform draw from the ball's
        distinct members (host-side; an option of the reference that CoDA never switches on)."""
        rows = idx.view(-1, self.nsample)
        counts = torch.zeros(rows.shape[0])
        for r, row in enumerate(rows):
            members = torch.unique(row)
            counts[r] = members.numel()
            refill = members[torch.randint(0, members.numel(), (self.nsample - members.numel(),), dtype=torch.long)]
            rows[r] = torch.cat((members, refill))
        return idx, counts.view(idx.shape[:2])

    def _grouped_coordinates(self, xyz, new_xyz):
        if not (self.sample_uniformly or xyz.requires_grad or new_xyz.requires_grad):
            idx, grouped = _ext.query_and_group_xyz(new_xyz, xyz, self.radius, self.nsample, self.normalize_xyz)
            return idx, grouped, None
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        counts = None
        if self.sample_uniformly:
            idx, counts = self._resample_uniformly(idx)
        grouped = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)  # (B,3,npoint,nsample)
        grouped = grouped - new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped = grouped / self.radius
        return idx, grouped, counts

    def forward(self, xyz, new_xyz, features=None):
        idx, grouped_xyz, unique_cnt = self._grouped_coordinates(xyz, new_xyz)
        if features is None:
            if not self.use_xyz:
                raise AssertionError("Cannot have not features and not use xyz as a feature!")
            new_features = grouped_xyz
        else:
            # (the masked encoder hands over a permuted view, which the operator wants dense)
            new_features = grouping_operation(features.contiguous(), idx)
            if self.use_xyz:
                new_features = torch.cat((grouped_xyz, new_features), dim=1)
        return _pack(new_features, (self.ret_grouped_xyz, grouped_xyz), (self.ret_unique_cnt, unique_cnt))


class GroupAll(nn.Module):
    """The degenerate grouper: the whole cloud is one group, (B,C,N) -> (B,3+C,1,N)."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz, self.ret_grouped_xyz = use_xyz, ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        coords = xyz.transpose(1, 2).unsqueeze(2)
        parts = ([coords] if (self.use_xyz or features is None) else []) + ([] if features is None else [features.unsqueeze(2)])
        new_features = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
        return _pack(new_features, (self.ret_grouped_xyz, coords))
