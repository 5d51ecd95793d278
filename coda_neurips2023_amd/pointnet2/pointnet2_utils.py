"""Autograd wrappers and grouping modules over ``pointnet2._ext``.

Mirror of third_party_pointnet2/pointnet2/pointnet2_utils.py:48-420: same
function names, argument order and differentiability.  The operators
themselves run in ``libcoda_hip.so`` (see ``_ext.py``).
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _ext


class FurthestPointSampling(Function):
    """xyz (B,N,3), npoint -> (B,npoint) int32; non-differentiable
    (pointnet2_utils.py:48-74)."""

    @staticmethod
    def forward(ctx, xyz, npoint):
        fps_inds = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(fps_inds)
        return fps_inds

    @staticmethod
    def backward(ctx, grad_out=None):
        return None, None


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

    def submit(self, point_clouds, module, wait_for="current"):
        dev = point_clouds.device
        if self._stream is None or self._stream.device != dev:
            self._stream = torch.cuda.Stream(device=dev)
        if isinstance(wait_for, torch.cuda.Event):
            self._stream.wait_event(wait_for)
        elif wait_for == "current":
            self._stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self._stream), torch.no_grad():
            xyz = point_clouds[..., 0:3].contiguous()
            prepared = module.prepare(xyz)
            if prepared is None:
                return False
            done = torch.cuda.Event()
            done.record(self._stream)
        self._pending.append((point_clouds, point_clouds._version, prepared, done))
        del self._pending[:-self._max]
        return True

    def take(self, point_clouds):
        """The prepared front of this very tensor (see ``PointnetSAModuleVotes.prepare``), or None."""
        for k, (pc, version, prepared, done) in enumerate(self._pending):
            if pc is point_clouds and version == point_clouds._version:
                del self._pending[k]
                if self.wait_for_counts:
                    done.synchronize()
                elif not done.query():
                    # the side stream is still busy (its workgroups need whole CUs and may have been
                    # starved): do not stall the host for the row count -- drop it (the shared MLP then
                    # runs on all rows) and let the compute stream wait on the device side
                    prepared = dict(prepared, total_host=None)
                cur = torch.cuda.current_stream(point_clouds.device)
                cur.wait_event(done)
                for v in prepared.values():
                    for t in (v if isinstance(v, tuple) else (v,)):
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(cur)
                return prepared
        return None


class GatherOperation(Function):
    """features (B,C,N), idx (B,npoint) -> (B,C,npoint) (pointnet2_utils.py:80-111)."""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.for_backwards = (idx, features.size(1), features.size(2))
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, _, n = ctx.for_backwards
        return _ext.gather_points_grad(grad_out.contiguous(), idx, n), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    """unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) L2, idx (B,n,3))
    (pointnet2_utils.py:117-145); the op returns squared distances, sqrt here."""

    @staticmethod
    def forward(ctx, unknown, known):
        dist2, idx = _ext.three_nn(unknown, known)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """features (B,c,m), idx (B,n,3), weight (B,n,3) -> (B,c,n)
    (pointnet2_utils.py:151-202)."""

    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.three_interpolate_for_backward = (idx, weight, features.size(2))
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        grad_features = _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, m)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)
    (pointnet2_utils.py:208-251)."""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.for_backwards = (idx, features.size(2))
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, n = ctx.for_backwards
        return _ext.group_points_grad(grad_out.contiguous(), idx, n), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """radius, nsample, xyz (B,N,3), new_xyz (B,npoint,3) -> (B,npoint,nsample)
    int32; non-differentiable (pointnet2_utils.py:257-285)."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        inds = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """Ball query + grouping (pointnet2_utils.py:291-373).

    When the coordinates carry no gradient (always the case for an input
    cloud) the xyz branch runs as ONE fused kernel that emits ``idx`` and the
    centred / normalised ``grouped_xyz`` directly; otherwise the reference's
    op-by-op sequence is kept so autograd sees the same graph.
    """

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False,
                 normalize_xyz=False, sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if self.ret_unique_cnt:
            assert self.sample_uniformly

    def _resample_uniformly(self, idx):
        # pointnet2_utils.py:333-342 -- host loop; never enabled by CoDA.
        unique_cnt = torch.zeros((idx.shape[0], idx.shape[1]))
        for i_batch in range(idx.shape[0]):
            for i_region in range(idx.shape[1]):
                unique_ind = torch.unique(idx[i_batch, i_region, :])
                num_unique = unique_ind.shape[0]
                unique_cnt[i_batch, i_region] = num_unique
                sample_ind = torch.randint(0, num_unique, (self.nsample - num_unique,),
                                           dtype=torch.long)
                all_ind = torch.cat((unique_ind, unique_ind[sample_ind]))
                idx[i_batch, i_region, :] = all_ind
        return idx, unique_cnt

    def forward(self, xyz, new_xyz, features=None):
        fused = not (self.sample_uniformly or xyz.requires_grad or new_xyz.requires_grad)
        unique_cnt = None
        if fused:
            idx, grouped_xyz = _ext.query_and_group_xyz(new_xyz, xyz, self.radius, self.nsample,
                                                        self.normalize_xyz)
        else:
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
            if self.sample_uniformly:
                idx, unique_cnt = self._resample_uniformly(idx)
            xyz_trans = xyz.transpose(1, 2).contiguous()
            grouped_xyz = grouping_operation(xyz_trans, idx)  # (B,3,npoint,nsample)
            grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
            if self.normalize_xyz:
                grouped_xyz /= self.radius

        if features is not None:
            # the masked encoder hands over a permuted view (transformer.py:199-201); the
            # reference op would assert on it, here it is made dense
            grouped_features = grouping_operation(features.contiguous(), idx)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1)
            else:
                new_features = grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz

        ret = [new_features]
        if self.ret_grouped_xyz:
            ret.append(grouped_xyz)
        if self.ret_unique_cnt:
            ret.append(unique_cnt)
        return ret[0] if len(ret) == 1 else tuple(ret)


class GroupAll(nn.Module):
    """Group every point into one region (pointnet2_utils.py:376-420)."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1)
            else:
                new_features = grouped_features
        else:
            new_features = grouped_xyz
        if self.ret_grouped_xyz:
            return new_features, grouped_xyz
        return new_features
