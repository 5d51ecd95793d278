"""CPU execution of the hot path for the reported baseline -- TEST INFRASTRUCTURE.

``patched()`` temporarily replaces the two kernel seams of the product package
(``pointnet2._ext`` operators and ``attention_core.attention``) with the CPU
oracle (C restatement of the ops, plain torch fp32 attention), so that the SAME
host-side module graph can be timed / checked on the host cores.  Only
``bench.py``'s cpu_baseline leg and ``tests/`` use it; the product never imports
this module and has no CPU path of its own.
"""
import contextlib

import torch

from . import pointnet2_oracle as O


def attention_ref(q, k, v, mask, scale, dropout_p, need_weights):
    """Plain fp32 softmax(q*scale @ k^T + mask) [dropout] @ v on (L,B,h,d) inputs."""
    qh, kh, vh = (t.permute(1, 2, 0, 3) for t in (q, k, v))
    scores = torch.matmul(qh * scale, kh.transpose(-1, -2))
    if mask is not None:
        scores = scores.masked_fill(mask, float("-inf"))
    probs = torch.softmax(scores, dim=-1)
    if dropout_p > 0.0:
        probs = torch.nn.functional.dropout(probs, p=dropout_p)
    out = torch.matmul(probs, vh)
    return out.permute(2, 0, 1, 3), (probs if need_weights else None)


def _bf16(t):
    """round-to-nearest-even to bfloat16 (8 significand bits), kept in the tensor's own dtype"""
    return t.to(torch.bfloat16).to(t.dtype)


class _AttentionBf16(torch.autograd.Function):
    """The attention core as the bf16-MFMA kernels compute it (csrc/attention_bf16.hip; include/coda_attention.h,
    MFMA operand type 1): EVERY operand of a matrix product is rounded to bfloat16 -- Q (pre-multiplied by
    scale * log2 e, the kernels work in log2 units), K, V, dO, the un-normalised probabilities and dS -- while the
    products accumulate, and soft-max / lse / delta are evaluated, in the tensors' precision (float32; float64 for a
    judge).  A checker for configs[4]: against a plain fp32 reference the bf16 mode can only be held at 2e-2 (core) /
    1e-1 (step gradients), which would hide a kernel bug of that size; against THIS function what remains is the
    difference between two placements of the same roundings (running vs final row maximum)."""

    @staticmethod
    def forward(ctx, q, k, v, mask, scale):
        qh, kh, vh = (t.permute(1, 2, 0, 3) for t in (q, k, v))        # (B,h,L,d)
        log2e = 1.4426950408889634
        qs, kb, vb = _bf16(qh * (scale * log2e)), _bf16(kh), _bf16(vh)
        s2 = torch.matmul(qs, kb.transpose(-1, -2))                    # scores in log2 units
        if mask is not None:
            s2 = s2.masked_fill(mask, float("-inf"))
        m = s2.amax(dim=-1, keepdim=True)
        m = torch.where(torch.isinf(m), torch.zeros_like(m), m)        # fully masked rows -> zeros, as the kernels
        pt = torch.exp2(s2 - m)                                        # un-normalised probabilities
        lsum = pt.sum(dim=-1, keepdim=True)
        inv = torch.where(lsum > 0, 1.0 / lsum, torch.zeros_like(lsum))
        out = torch.matmul(_bf16(pt), vb) * inv
        lse2 = m + torch.log2(torch.where(lsum > 0, lsum, torch.ones_like(lsum)))  # log-sum-exp in log2 units
        dead = (lsum <= 0) if mask is None else ((lsum <= 0) | mask)
        ctx.save_for_backward(qh, kb, vb, pt * inv, out, lse2, dead.expand_as(pt))
        ctx.scale = scale
        return out.permute(2, 0, 1, 3)

    @staticmethod
    def backward(ctx, dout):
        """Two kernels, two recomputations of P (csrc/attention_bf16.hip): the dQ kernel recomputes the scores like the
        forward (Q pre-scaled, then rounded), the dK/dV kernel from the UNSCALED rounded Q with the scale applied in
        fp32 -- both against the forward's lse, so the dK/dV kernel's probabilities are the forward's up to the
        difference of those two roundings of Q."""
        qh, kb, vb, probs, out, lse2, dead = ctx.saved_tensors
        log2e = 1.4426950408889634
        do = dout.permute(1, 2, 0, 3)
        dob = _bf16(do)
        qb = _bf16(qh)
        delta = (do * out).sum(dim=-1, keepdim=True)                   # rowsum(dO * O): not a matrix product
        dp = torch.matmul(dob, vb.transpose(-1, -2))
        # dK / dV kernel
        p_kv = torch.exp2(torch.matmul(qb, kb.transpose(-1, -2)) * (ctx.scale * log2e) - lse2)
        p_kv = torch.where(dead, torch.zeros_like(p_kv), p_kv)
        dv = torch.matmul(_bf16(p_kv).transpose(-1, -2), dob)
        dk = torch.matmul(_bf16(p_kv * (dp - delta)).transpose(-1, -2), qb) * ctx.scale
        # dQ kernel
        dq = torch.matmul(_bf16(probs * (dp - delta)), kb) * ctx.scale
        return dq.permute(2, 0, 1, 3), dk.permute(2, 0, 1, 3), dv.permute(2, 0, 1, 3), None, None


def attention_ref_bf16(q, k, v, mask, scale, dropout_p, need_weights):
    """``attention_ref`` with the operand roundings of the bf16-MFMA mode (dropout-free: the parity tests of that mode
    run in eval mode / with p = 0)."""
    if dropout_p > 0.0:
        raise NotImplementedError("the bf16 oracle is dropout-free")
    out = _AttentionBf16.apply(q, k, v, mask, scale)
    probs = None
    if need_weights:
        probs = attention_ref(q, k, v, mask, scale, 0.0, True)[1]
    return out, probs


def generalized_box3d_iou(corners1, corners2, nums_k2, rotated_boxes=True, return_inter_vols_only=False,
                          needs_grad=False):
    """box_util.generalized_box3d_iou's signature on CPU tensors, from oracle/box_giou_oracle.c."""
    from . import box_giou_oracle as BO
    assert not needs_grad
    out = BO.generalized_box3d_iou_c(corners1.detach().numpy(), corners2.detach().numpy(),
                                     None if nums_k2 is None else nums_k2.numpy(), bool(rotated_boxes),
                                     return_inter_vols_only)
    return torch.from_numpy(out)


class _AnyDtypeExt:
    """The nine operators for tensors of any floating dtype -- the float64 JUDGE of the full-size parity tests
    (tests/test_full_step_gpu.py): index-producing operators run the C oracle on the float32 image of their input
    (the clouds ARE float32 data, so the cast back is lossless and the indices are the float32 path's), gathers and
    their adjoints are plain torch indexing in the tensor's own dtype."""

    def __init__(self):
        self._f32 = O.TorchExt()

    def furthest_point_sampling(self, points, nsamples):
        return self._f32.furthest_point_sampling(points.detach().float(), nsamples)

    def ball_query(self, new_xyz, xyz, radius, nsample):
        return self._f32.ball_query(new_xyz.detach().float(), xyz.detach().float(), radius, nsample)

    def three_nn(self, unknowns, knows):
        d, i = self._f32.three_nn(unknowns.detach().float(), knows.detach().float())
        return d.to(unknowns.dtype), i

    @staticmethod
    def gather_points(points, idx):                       # (B,C,N), (B,M) -> (B,C,M)
        return torch.gather(points, 2, idx.long().unsqueeze(1).expand(-1, points.shape[1], -1))

    @staticmethod
    def gather_points_grad(grad_out, idx, n):             # (B,C,M) -> (B,C,n)
        out = grad_out.new_zeros(grad_out.shape[0], grad_out.shape[1], n)
        return out.scatter_add_(2, idx.long().unsqueeze(1).expand(-1, grad_out.shape[1], -1), grad_out)

    @staticmethod
    def group_points(points, idx):                        # (B,C,N), (B,M,S) -> (B,C,M,S)
        b, c, _ = points.shape
        flat = idx.long().reshape(b, 1, -1).expand(-1, c, -1)
        return torch.gather(points, 2, flat).view(b, c, idx.shape[1], idx.shape[2])

    @staticmethod
    def group_points_grad(grad_out, idx, n):              # (B,C,M,S) -> (B,C,n)
        b, c = grad_out.shape[:2]
        out = grad_out.new_zeros(b, c, n)
        return out.scatter_add_(2, idx.long().reshape(b, 1, -1).expand(-1, c, -1), grad_out.reshape(b, c, -1))

    @staticmethod
    def three_interpolate(points, idx, weight):           # (B,C,m), (B,n,3), (B,n,3) -> (B,C,n)
        b, c, _ = points.shape
        g = torch.gather(points, 2, idx.long().reshape(b, 1, -1).expand(-1, c, -1)).view(b, c, idx.shape[1], 3)
        return (g * weight.to(points.dtype).unsqueeze(1)).sum(-1)

    @staticmethod
    def three_interpolate_grad(grad_out, idx, weight, m):  # (B,C,n) -> (B,C,m)
        b, c, n = grad_out.shape
        contrib = (grad_out.unsqueeze(-1) * weight.to(grad_out.dtype).unsqueeze(1)).reshape(b, c, n * 3)
        out = grad_out.new_zeros(b, c, m)
        return out.scatter_add_(2, idx.long().reshape(b, 1, -1).expand(-1, c, -1), contrib)


@contextlib.contextmanager
def patched(any_dtype=False, attention="fp32"):
    """``any_dtype=True``: operators that follow the tensors' dtype (``_AnyDtypeExt``), for a float64 run of the
    same module graph as the judge between two float32 evaluations.  ``attention="bf16"``: the attention seam gets
    ``attention_ref_bf16`` (the operand roundings of the bf16-MFMA mode, configs[4])."""
    from coda_neurips2023_amd import attention_core
    from coda_neurips2023_amd.pointnet2 import _ext

    ext = _AnyDtypeExt() if any_dtype else O.TorchExt()
    names = ["gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn",
             "three_interpolate", "three_interpolate_grad", "ball_query", "group_points",
             "group_points_grad"]
    saved = {n: getattr(_ext, n) for n in names + ["query_and_group_xyz"]}
    saved_attn = attention_core.attention

    def query_and_group_xyz(new_xyz, xyz, radius, nsample, normalize_xyz):
        idx = ext.ball_query(new_xyz, xyz, radius, nsample)
        grouped = ext.group_points(xyz.transpose(1, 2).contiguous(), idx)
        grouped = grouped - new_xyz.transpose(1, 2).unsqueeze(-1)
        if normalize_xyz:
            grouped = grouped / radius
        return idx, grouped

    try:
        for n in names:
            setattr(_ext, n, getattr(ext, n))
        _ext.query_and_group_xyz = query_and_group_xyz
        attention_core.attention = attention_ref_bf16 if attention == "bf16" else attention_ref
        yield
    finally:
        for n, f in saved.items():
            setattr(_ext, n, f)
        attention_core.attention = saved_attn
