"""Fused shared-MLP + max-pool of the set-abstraction layer (xyz-only input).

Computes exactly what ``SharedMLP`` (1x1 Conv2d, no bias -> BatchNorm2d -> ReLU, per
layer) followed by ``F.max_pool2d`` over the nsample axis computes in the reference
(pointnet2_modules.py:247-253, pytorch_utils.py:8-117), including train-mode batch
statistics, running-statistics updates and SyncBatchNorm semantics, but on
channels-last activations.  Two implementations:

* ``_MfmaMlpPool`` (default for the pre-encoder's widths [3, 64, 128, 256]): the 1x1 convolutions
  are hand-written fp32-MFMA GEMMs with BN / ReLU / statistics / pooling fused into their
  prologues and epilogues (``csrc/sa_mfma.hip``, include/coda_sa_mlp.h "MFMA pipeline"); the
  de-duplicated groups are packed on the device, so nothing waits for a row count on the host;
* ``_FusedMlpPool`` (other widths; ``CODA_SA_MLP=streams`` for the A/B): the streaming kernels of
  ``csrc/sa_mlp.hip`` around plain library GEMMs (``torch.mm`` -> rocBLAS).

The parameters stay in the reference modules (``mlp_module.layer{i}.conv.weight``,
``...bn.bn.*``); this file only holds the autograd functions that read them.
"""
import os

import torch
import torch.distributed as dist

from .. import _lib
from ..linear_fn import tn_gemm


def _p(t):
    return t.data_ptr() if t is not None else None


def _stream():
    return _lib.current_stream_handle()


def _call(name, *args):
    lib = _lib.load()
    _lib.check(getattr(lib, name)(*args, _stream()), name)


# measurement aid (bench.py's roofline figures of the MFMA kernels): HIP events on the launch stream around each call
_TIMING = None
# dev probe: workgroups of the two forward MFMA launches (0: the library's 2 per CU); see DESIGN.md section 7, round 6
_FWD_BLOCKS = int(os.environ.get("CODA_SA_FWD_BLOCKS", "0"))
_LEAVE_SAMPLING_CUS = os.environ.get("CODA_SA_LEAVE_SAMPLING_CUS", "1") != "0"  # A/B


def enable_kernel_timing():
    """{label: [(start event, end event), ...]} filled by the MFMA pipeline's GEMM launches until disable_kernel_timing()."""
    global _TIMING
    _TIMING = {}
    return _TIMING


def disable_kernel_timing():
    global _TIMING
    _TIMING = None


def _call_timed(label, name, *args):
    if _TIMING is None:
        return _call(name, *args)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    _call(name, *args)
    end.record()
    _TIMING.setdefault(label, []).append((start, end))


def _is_sync(bn):
    return isinstance(bn, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized() \
        and dist.get_world_size(bn.process_group) > 1


def _all_reduce(t, bn):
    if _is_sync(bn):
        dist.all_reduce(t, group=bn.process_group)
    return t


_ROW_PAD = max(2048, int(os.environ.get("CODA_SA_ROW_PAD", "16384")) // 2048 * 2048)


def count_distinct_rows(idx):
    """idx (B,M,S) int32 ball-query indices -> (cnt (G,), goff (G+1,) int64, total (1,) int64):
    distinct rows per group, their exclusive prefix sum and the overall count.  ball_query fills
    the slots a group has no point for with copies of its FIRST hit (ball_query_gpu.cu:35-48;
    real hits are distinct ascending indices)."""
    b, m, s = idx.shape
    idx2 = idx.view(b * m, s)
    cnt = (idx2[:, 1:] != idx2[:, :1]).sum(1) + 1
    goff = torch.zeros(b * m + 1, dtype=torch.int64, device=idx.device)
    torch.cumsum(cnt, 0, out=goff[1:])
    return cnt, goff, goff[-1:]


def compact_groups(idx, grouped_cl, counts=None, total=None, min_saving=0.25):
    """De-duplicate the padded groups of a ball query.

    idx (B,M,S) int32, grouped_cl (B,M,S,3).  Identical rows stay identical through the whole
    shared MLP, so only the distinct rows of a group need to be computed.  Returns
    ``(x (Pp,3), row_weight (Pp,), group_offsets (G+1,) int32)`` with Pp the number of distinct
    rows rounded up to a multiple of ``_ROW_PAD`` (zero rows of weight 0), or None when fewer than
    ``min_saving`` of the rows are copies.  The row count sizes the GEMMs, so it must be known
    on the host: pass ``counts`` / ``total`` from an earlier ``count_distinct_rows`` whose
    result has already been copied back (the sampling prefetcher does), otherwise this call
    synchronises."""
    b, m, s = idx.shape
    g = b * m
    dev = idx.device
    cnt, goff, tot = counts if counts is not None else count_distinct_rows(idx)
    if total is None:
        total = int(tot.item())
    if total > (1.0 - min_saving) * g * s:
        return None
    # rows rounded up to _ROW_PAD (zero rows of weight 0): the row count is data-dependent, and a coarse grid keeps
    # the set of GEMM shapes small enough for a table of tuned library kernels (tuning.py) at 2.5 % more rows
    pp = -(-total // _ROW_PAD) * _ROW_PAD
    if pp >= g * s:  # small inputs: the padded compact form would not be smaller than the padded groups themselves
        return None
    # one launch (coda_sa_compact_groups_f32): a group's distinct rows are its first cnt slots
    x = torch.empty((pp, 3), dtype=torch.float32, device=dev)
    roww = torch.empty(pp, dtype=torch.float32, device=dev)
    goff32 = torch.empty(g + 1, dtype=torch.int32, device=dev)
    src = grouped_cl if grouped_cl.is_contiguous() else grouped_cl.contiguous()
    _call("coda_sa_compact_groups_f32", _p(src), _p(cnt.contiguous()), _p(goff), _p(x), _p(roww), _p(goff32), g, s, total,
          pp)
    return x, roww, goff32


def _bwd_coef(ctx, i, sums, st, gamma, bn, training, n, layout, grads, dev):
    """d beta / d gamma of layer i from the local sums, then (after SyncBN's all-reduce) the coefficient table of
    the layer's backward kernel -- one launch (two with SyncBN across ranks)."""
    c = st.shape[1]
    coef = torch.empty((5 if layout else 7, c), dtype=torch.float32, device=dev)
    dbeta, dgamma = torch.empty(c, dtype=torch.float32, device=dev), torch.empty(c, dtype=torch.float32, device=dev)
    grads[3 * i + 2], grads[3 * i + 1] = dbeta, dgamma
    if training and _is_sync(bn):
        _call("coda_sa_bn_bwd_coef_f32", _p(sums), 0.0, None, None, None, layout, _p(dbeta), _p(dgamma), c)
        _all_reduce(sums, bn)
        _call("coda_sa_bn_bwd_coef_f32", _p(sums), float(n), _p(gamma), _p(st), _p(coef), layout, None, None, c)
    else:
        _call("coda_sa_bn_bwd_coef_f32", _p(sums), float(n) if training else 0.0, _p(gamma), _p(st), _p(coef), layout,
              _p(dbeta), _p(dgamma), c)
    return coef


class _FusedMlpPool(torch.autograd.Function):
    """x (P,3) grouped xyz channels-last, P = groups * nsample (or the distinct rows of
    ``compact_groups`` with ``dedup = (row_weight, group_offsets)``).  Returns (groups, C_last)."""

    @staticmethod
    def forward(ctx, x, groups, nsample, bns, training, dedup, *params):
        # params: for each layer: conv weight (Cout, Cin[,1,1]), bn weight, bn bias
        nl = len(bns)
        dev = x.device
        p = x.shape[0]
        roww, goff = dedup if dedup is not None else (None, None)
        n_rows = groups * nsample  # rows the statistics are taken over (copies included)
        ws = [params[3 * i].reshape(params[3 * i].shape[0], -1) for i in range(nl)]
        gammas = [params[3 * i + 1] for i in range(nl)]
        betas = [params[3 * i + 2] for i in range(nl)]
        world = [dist.get_world_size(bn.process_group) if _is_sync(bn) else 1 for bn in bns]

        saved_pre, saved_act, stats = [], [], []
        sums = torch.empty(2 * max(w.shape[0] for w in ws), dtype=torch.float64, device=dev)
        cur = x  # input of layer i (post-activation of layer i-1)
        pool = None
        for i in range(nl):
            c = ws[i].shape[0]
            last = i == nl - 1
            w1 = ws[0].contiguous() if i == 0 else None
            if i == 0:
                pre = None  # recomputed on the fly from x
                src = x
            else:
                pre = torch.mm(cur, ws[i].t())  # (P, Cin) @ (Cin, Cout): plain library GEMM
                src = pre
            s = sums[:2 * c]
            if last:
                ymax = torch.empty((groups, c), dtype=torch.float32, device=dev)
                ymin = torch.empty_like(ymax)
                amax = torch.empty((groups, c), dtype=torch.int32, device=dev)
                amin = torch.empty_like(amax)
                if i == 0:
                    raise RuntimeError("fused SA MLP needs at least two layers")
                _call("coda_sa_col_stats_pool_f32", _p(src), groups, nsample, c, _p(roww), _p(goff), _p(s), _p(ymax),
                      _p(ymin), _p(amax), _p(amin))
                pool = (ymax, ymin, amax, amin)
            elif training:
                _call("coda_sa_col_stats_f32", _p(src), _p(w1), p, c, _p(roww), _p(s))
            bn = bns[i]
            st = torch.empty((4, c), dtype=torch.float32, device=dev)  # scale, shift, mean, invstd
            if training and (bn.momentum is not None or not bn.track_running_stats or bn.running_mean is None):
                # one launch: statistics -> scale / shift and the running-statistics update
                _all_reduce(s, bn)
                track = bn.track_running_stats and bn.running_mean is not None
                _call("coda_sa_bn_finalize_f32", _p(s), float(n_rows * world[i]), float(bn.eps),
                      float(bn.momentum) if track else 0.0, _p(gammas[i].detach()), _p(betas[i].detach()),
                      _p(bn.running_mean) if track else None, _p(bn.running_var) if track else None,
                      _p(bn.num_batches_tracked) if track else None, _p(st), c)
            elif training:  # cumulative moving average (momentum=None): the factor depends on a device counter
                tot = _all_reduce(s.clone(), bn)
                n = float(n_rows * world[i])
                mean = tot[:c] / n
                var = (tot[c:] / n - mean * mean).clamp_(min=0.0)
                invstd = torch.rsqrt(var + bn.eps)
                with torch.no_grad():
                    mom = 1.0 / float(bn.num_batches_tracked + 1)
                    bn.running_mean.mul_(1 - mom).add_(mean.to(torch.float32), alpha=mom)
                    bn.running_var.mul_(1 - mom).add_((var * (n / max(n - 1.0, 1.0))).to(torch.float32), alpha=mom)
                    bn.num_batches_tracked += 1
                st[2], st[3] = mean.to(torch.float32), invstd.to(torch.float32)
                st[0] = gammas[i] * st[3]
                st[1] = betas[i] - st[2] * st[0]
            else:
                st[2], st[3] = bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)
                st[0] = gammas[i] * st[3]
                st[1] = betas[i] - st[2] * st[0]
            scale, shift = st[0], st[1]
            stats.append(st)
            saved_pre.append(pre)
            if not last:
                act = torch.empty((p, c), dtype=torch.float32, device=dev)
                _call("coda_sa_bn_relu_apply_f32", _p(src), _p(w1), _p(scale), _p(shift), p, c, _p(act))
                saved_act.append(act)
                cur = act

        ymax, ymin, amax, amin = pool
        ysel, sel, out = torch.empty_like(ymax), torch.empty_like(amax), torch.empty_like(ymax)
        _call("coda_sa_pool_select_f32", _p(ymax), _p(ymin), _p(amax), _p(amin), _p(stats[-1]), _p(ysel), _p(sel), _p(out),
              groups, ws[-1].shape[0])             # BN is monotone per channel: pool the pre-BN values

        ctx.meta = (groups, nsample, bns, training, nl, world, n_rows)
        ctx.dedup = dedup
        ctx.wshape = [params[3 * i].shape for i in range(nl)]
        ctx.stats = stats
        ctx.save_for_backward(x, ysel, sel, out, *[t for t in saved_pre if t is not None], *saved_act, *ws, *gammas)
        ctx.n_pre = sum(t is not None for t in saved_pre)
        return out

    @staticmethod
    def backward(ctx, gout):
        groups, nsample, bns, training, nl, world, n_rows = ctx.meta
        roww, goff = ctx.dedup if ctx.dedup is not None else (None, None)
        saved = ctx.saved_tensors
        x, ysel, sel, out = saved[:4]
        pres = [None] + list(saved[4:4 + ctx.n_pre])              # pre-BN activations of layers 1..nl-1
        acts = list(saved[4 + ctx.n_pre:4 + ctx.n_pre + nl - 1])  # post-activation of layers 0..nl-2
        ws = list(saved[4 + ctx.n_pre + nl - 1:4 + ctx.n_pre + 2 * nl - 1])
        gammas = list(saved[4 + ctx.n_pre + 2 * nl - 1:])
        dev = x.device
        p = x.shape[0]
        grads = [None] * (3 * nl)

        # ---- last layer: max-pool + ReLU + BN backward
        i = nl - 1
        c = ws[i].shape[0]
        st = ctx.stats[i]
        d = torch.empty_like(out)                                # (groups, C) at sample sel
        sums = torch.empty(2 * c, dtype=torch.float64, device=dev)
        _call("coda_sa_pool_bwd_stats_f32", _p(gout.contiguous()), _p(out), _p(ysel), _p(st), _p(d), groups, c, _p(sums))
        coef = _bwd_coef(ctx, i, sums, st, gammas[i], bns[i], training, n_rows * world[i], 1, grads, dev)
        # de-duplicated rows: the padding rows past the last group belong to no group and stay zero
        dy = (torch.zeros if roww is not None else torch.empty)((p, c), dtype=torch.float32, device=dev)
        _call("coda_sa_bn_bwd_sparse_f32", _p(pres[i]), _p(d), _p(sel.contiguous()), _p(coef), groups, nsample, c,
              _p(roww), _p(goff), _p(dy))

        # ---- hidden layers, top down
        while True:
            a_in = acts[i - 1]                                    # input of layer i
            grads[3 * i] = tn_gemm(dy, a_in).reshape(ctx.wshape[i])        # dW_i = dY^T A_{i-1} (split-K)
            da = torch.mm(dy, ws[i])                              # dA_{i-1} = dY W_i
            del dy
            i -= 1
            c = ws[i].shape[0]
            st = ctx.stats[i]
            first = i == 0
            src = x if first else pres[i]
            w1 = ws[0].contiguous() if first else None
            sums = torch.empty(2 * c, dtype=torch.float64, device=dev)
            _call("coda_sa_relu_bn_bwd_stats_f32", _p(da), _p(src), _p(w1), _p(st), p, c, _p(sums))
            prm7 = _bwd_coef(ctx, i, sums, st, gammas[i], bns[i], training, n_rows * world[i], 0, grads, dev)
            if first:
                dw1 = torch.empty(3 * c, dtype=torch.float64, device=dev)
                _call("coda_sa_relu_bn_bwd_apply_f32", _p(da), _p(src), _p(w1), _p(prm7), p, c, _p(roww), None, _p(dw1))
                grads[0] = dw1.view(3, c).t().to(torch.float32).reshape(ctx.wshape[0])
                break
            _call("coda_sa_relu_bn_bwd_apply_f32", _p(da), _p(src), None, _p(prm7), p, c, _p(roww), _p(da), None)
            dy = da
        return (None, None, None, None, None, None, *grads)



def _bn_stats(bn, s, n, gamma, beta, training, c, dev):
    """stats = [scale, shift, mean, invstd][c] of one BatchNorm layer from sums s = [sum y, sum y^2] over n rows
    (train mode: batch statistics + running-statistics update; SyncBatchNorm: sums all-reduced first) or from the
    running statistics (eval mode) -- the bookkeeping of ``_FusedMlpPool.forward``, shared with the MFMA path."""
    st = torch.empty((4, c), dtype=torch.float32, device=dev)
    if training and (bn.momentum is not None or not bn.track_running_stats or bn.running_mean is None):
        _all_reduce(s, bn)
        track = bn.track_running_stats and bn.running_mean is not None
        _call("coda_sa_bn_finalize_f32", _p(s), float(n), float(bn.eps), float(bn.momentum) if track else 0.0,
              _p(gamma.detach()), _p(beta.detach()), _p(bn.running_mean) if track else None,
              _p(bn.running_var) if track else None, _p(bn.num_batches_tracked) if track else None, _p(st), c)
    elif training:  # cumulative moving average (momentum=None): the factor depends on a device counter
        tot = _all_reduce(s.clone(), bn)
        mean = tot[:c] / n
        var = (tot[c:] / n - mean * mean).clamp_(min=0.0)
        invstd = torch.rsqrt(var + bn.eps)
        with torch.no_grad():
            mom = 1.0 / float(bn.num_batches_tracked + 1)
            bn.running_mean.mul_(1 - mom).add_(mean.to(torch.float32), alpha=mom)
            bn.running_var.mul_(1 - mom).add_((var * (n / max(n - 1.0, 1.0))).to(torch.float32), alpha=mom)
            bn.num_batches_tracked += 1
        st[2], st[3] = mean.to(torch.float32), invstd.to(torch.float32)
        st[0] = gamma * st[3]
        st[1] = beta - st[2] * st[0]
    else:
        st[2], st[3] = bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)
        st[0] = gamma * st[3]
        st[1] = beta - st[2] * st[0]
    return st


def pack_groups(idx, grouped_cl, widths, dedup=True):
    """The parameter-free front of the MFMA path (``coda_sa_pack_groups_f32``; the sampling prefetcher runs it on
    its side stream): the distinct rows of every ball-query group packed back to back, with their multiplicity,
    group offsets, (group, row-in-group) words and the 3x3 moments of the packed xyz -- all on the device, no row
    count travels to the host.  Also zeroes the statistics accumulators of the two MFMA layers -- ONCE: the last element
    of the result is a use counter, and a second forward on the same tuple (a prefetched front used twice, an
    activation-checkpoint recompute) zeroes them itself instead of adding into the first pass's -- possibly already
    all-reduced -- sums.  idx (B,M,S) int32, grouped_cl (B,M,S,3) / (B*M*S,3) float32 -> tuple of tensors + counter.
    Capacity note: x / y2 / y3 of the forward are sized for B*M*S rows (the packed count is not known on the host):
    1.5 GB at 8 x 2048 x 64, of which the de-duplicated rows (30 % on the bench scenes) are touched."""
    b, m, s = idx.shape
    g = b * m
    dev = idx.device
    cap = g * s
    f32 = dict(dtype=torch.float32, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    x = torch.empty((cap, 3), **f32)
    roww = torch.empty(cap, **f32)
    goff = torch.empty(g + 1, **i32)
    grow = torch.empty(cap, **i32)
    counts = torch.empty(max(g, 1), **i32)
    mom = torch.empty(10, dtype=torch.float64, device=dev)
    sums = torch.empty(2 * sum(widths), dtype=torch.float64, device=dev)  # [l1 | l2 | l3] x [sum, sum of squares]
    zero = sums[2 * widths[0]:]
    src = grouped_cl if grouped_cl.is_contiguous() else grouped_cl.contiguous()
    _call("coda_sa_pack_groups_f32", _p(src), _p(idx.contiguous()), 1 if dedup else 0, _p(x), _p(roww), _p(goff), _p(grow),
          _p(mom), _p(counts), _p(zero), zero.numel(), g, s)
    return x, roww, goff, grow, mom, sums, {"uses": 0}


class _MfmaMlpPool(torch.autograd.Function):
    """``packed`` = pack_groups(...) of the grouped xyz.  Returns (groups, C3).  Three layers [3 -> C1 -> C2 -> C3]
    (coda_sa_mfma_supported)."""

    @staticmethod
    def forward(ctx, packed, groups, nsample, bns, training, *params):
        lib = _lib.load()
        x, roww, goff, grow, mom, sums, state = packed
        dev = x.device
        cap = x.shape[0]
        c_first = params[0].shape[0]
        if state["uses"] > 0:  # the accumulators hold an earlier pass's sums: this pass starts from zero like the first
            sums[2 * c_first:].zero_()
        state["uses"] += 1
        ws = [params[3 * i].reshape(params[3 * i].shape[0], -1).contiguous() for i in range(3)]
        gammas = [params[3 * i + 1] for i in range(3)]
        betas = [params[3 * i + 2] for i in range(3)]
        c1, c2, c3 = (w.shape[0] for w in ws)
        world = [dist.get_world_size(bn.process_group) if _is_sync(bn) else 1 for bn in bns]
        n_rows = groups * nsample  # rows the statistics are taken over (copies included)
        nblk = _FWD_BLOCKS or lib.coda_sa_mfma_blocks(0)
        if not _FWD_BLOCKS and _LEAVE_SAMPLING_CUS:
            # the next batch's sampling front may be running on the side stream and holds whole CUs: two workgroups for
            # each FREE CU (any count is correct -- the kernels split the rows evenly over their grid)
            from .pointnet2_utils import sampling_busy_cus
            busy = sampling_busy_cus()
            if busy:
                nblk = max(nblk // 2, nblk - 2 * busy)
        f32 = dict(dtype=torch.float32, device=dev)
        s1, s2, s3 = sums[:2 * c1], sums[2 * c1:2 * (c1 + c2)], sums[2 * (c1 + c2):]

        # layer 1: y1 = x W1^T is linear in x -> batch statistics from the 3x3 moments; never formed in memory
        if training:
            _call("coda_sa_l1_sums_f32", _p(mom), _p(ws[0]), _p(s1), c1)
        st1 = _bn_stats(bns[0], s1, float(n_rows * world[0]), gammas[0], betas[0], training, c1, dev)
        # layer 2: prologue = layer 1 recomputed from x + BN + ReLU, epilogue = statistics
        y2 = torch.empty((cap, c2), **f32)
        _call_timed("fwd2", "coda_sa_mfma_fwd_f32", _p(x), _p(ws[0]), _p(st1), _p(ws[1]), _p(roww), _p(goff), _p(grow), groups, nsample,
              c1, c2, _p(y2), _p(s2), None, None, None, None, None, None, nblk)
        st2 = _bn_stats(bns[1], s2, float(n_rows * world[1]), gammas[1], betas[1], training, c2, dev)
        # layer 3: prologue = BN + ReLU of layer 2, epilogue = statistics + pooling (sign of gamma: max or min)
        y3 = torch.empty((cap, c3), **f32)
        ysel = torch.empty((groups, c3), **f32)
        sel = torch.empty((groups, c3), dtype=torch.int32, device=dev)
        g3 = gammas[2].detach()
        out = torch.empty((groups, c3), **f32)
        part_y = torch.empty((nblk, c3), **f32)
        part_sel = torch.empty((nblk, c3), dtype=torch.int32, device=dev)
        part_gid = torch.empty(nblk, dtype=torch.int32, device=dev)
        _call_timed("fwd3", "coda_sa_mfma_fwd_f32", _p(y2), None, _p(st2), _p(ws[2]), _p(roww), _p(goff), _p(grow), groups, nsample,
              c2, c3, _p(y3), _p(s3), _p(g3), _p(ysel), _p(sel), _p(part_y), _p(part_sel), _p(part_gid), nblk)
        st3 = _bn_stats(bns[2], s3, float(n_rows * world[2]), gammas[2], betas[2], training, c3, dev)
        _call("coda_sa_pool_finish_f32", _p(ysel), _p(sel), _p(part_y), _p(part_sel), _p(part_gid), _p(goff), _p(g3),
              _p(st3), _p(out), groups, c3, nblk)

        ctx.meta = (groups, nsample, bns, training, world, n_rows, nblk, lib.coda_sa_mfma_blocks(1))
        if _TIMING is not None:
            _TIMING.setdefault("rows", []).append(goff[-1:])  # packed row count of this call (read back after the run)
        ctx.wshape = [params[3 * i].shape for i in range(3)]
        ctx.stats = [st1, st2, st3]
        ctx.save_for_backward(x, roww, goff, grow, mom, y2, y3, ysel, sel, out, *ws, *gammas)
        return out

    @staticmethod
    def backward(ctx, gout):
        groups, nsample, bns, training, world, n_rows, nblk, nblk_w = ctx.meta
        x, roww, goff, grow, mom, y2, y3, ysel, sel, out = ctx.saved_tensors[:10]
        ws = list(ctx.saved_tensors[10:13])
        gammas = list(ctx.saved_tensors[13:16])
        st1, st2, st3 = ctx.stats
        dev = x.device
        cap = x.shape[0]
        c1, c2, c3 = (w.shape[0] for w in ws)
        grads = [None] * 9
        f32 = dict(dtype=torch.float32, device=dev)
        f64 = dict(dtype=torch.float64, device=dev)

        # ---- layer 3: max-pool + ReLU backward on the pooled tensor, then BN backward inside the two GEMM kernels
        d = torch.empty_like(out)
        sums = torch.empty(2 * c3, **f64)
        _call("coda_sa_pool_bwd_stats_f32", _p(gout.contiguous()), _p(out), _p(ysel), _p(st3), _p(d), groups, c3, _p(sums))
        coef3 = _bwd_coef(ctx, 2, sums, st3, gammas[2], bns[2], training, n_rows * world[2], 1, grads, dev)
        dmid2 = torch.empty((cap, c2), **f32)
        sums2 = torch.empty(2 * c2, **f64)
        _call_timed("dx3", "coda_sa_mfma_bwd_dx_f32", _p(y3), None, _p(d), _p(sel), _p(coef3), 1, _p(ws[2]), _p(y2), None, _p(st2),
              _p(roww), _p(goff), _p(grow), groups, nsample, c2, c3, _p(dmid2), _p(sums2), nblk)
        partials = torch.empty((nblk_w, c3 * c2), **f32)
        dw3 = torch.empty((c3, c2), **f32)
        _call_timed("dw3", "coda_sa_mfma_bwd_dw_f32", _p(y3), None, _p(d), _p(sel), _p(coef3), 1, _p(y2), None, _p(st2), _p(roww),
              _p(goff), _p(grow), groups, nsample, c2, c3, _p(partials), _p(dw3), nblk_w)
        grads[6] = dw3.reshape(ctx.wshape[2])

        # ---- layer 2
        prm2 = _bwd_coef(ctx, 1, sums2, st2, gammas[1], bns[1], training, n_rows * world[1], 0, grads, dev)
        sums1 = torch.empty(5 * c1, **f64)
        _call_timed("dx2", "coda_sa_mfma_bwd_dx_f32", _p(y2), _p(dmid2), None, None, _p(prm2), 0, _p(ws[1]), _p(x), _p(ws[0]), _p(st1),
              _p(roww), _p(goff), _p(grow), groups, nsample, c1, c2, None, _p(sums1), nblk)
        dw2 = torch.empty((c2, c1), **f32)
        _call_timed("dw2", "coda_sa_mfma_bwd_dw_f32", _p(y2), _p(dmid2), None, None, _p(prm2), 0, _p(x), _p(ws[0]), _p(st1), _p(roww),
              _p(goff), _p(grow), groups, nsample, c1, c2, _p(partials), _p(dw2), nblk_w)
        grads[3] = dw2.reshape(ctx.wshape[1])

        # ---- layer 1, closed form: dW1 / dgamma / dbeta from five sums per channel and the xyz moments
        sums_bn = sums1
        if training and _is_sync(bns[0]):
            sums_bn = sums1[:2 * c1].clone()
            _all_reduce(sums_bn, bns[0])
        dw1 = torch.empty((c1, 3), **f32)
        dbeta, dgamma = torch.empty(c1, **f32), torch.empty(c1, **f32)
        _call("coda_sa_l1_bwd_f32", _p(sums1), _p(sums_bn), float(n_rows * world[0]) if training else 0.0, _p(gammas[0]),
              _p(st1), _p(mom), _p(ws[0]), _p(dw1), _p(dbeta), _p(dgamma), c1)
        grads[0], grads[1], grads[2] = dw1.reshape(ctx.wshape[0]), dgamma, dbeta
        return (None, None, None, None, None, *grads)


def mfma_eligible(mlp_module, nsample):
    """Three layers at the widths csrc/sa_mfma.hip is instantiated for (the pre-encoder's [3, 64, 128, 256])."""
    if os.environ.get("CODA_SA_MLP", "mfma") != "mfma":
        return False
    layers = list(mlp_module.children())
    if len(layers) != 3:
        return False
    c = [layer.conv.out_channels for layer in layers]
    return bool(_lib.load().coda_sa_mfma_supported(c[0], c[1], c[2], int(nsample)))


def mfma_widths(mlp_module):
    return [layer.conv.out_channels for layer in mlp_module.children()]


def mfma_mlp_pool(packed, groups, nsample, mlp_module):
    """packed = pack_groups(idx, grouped, mfma_widths(mlp_module)) -> (groups, C3)."""
    layers = list(mlp_module.children())
    bns = [layer.bn.bn for layer in layers]
    params = []
    for layer in layers:
        params += [layer.conv.weight, layer.bn.bn.weight, layer.bn.bn.bias]
    return _MfmaMlpPool.apply(packed, groups, nsample, bns, bns[0].training, *params)


def fused_mlp_pool(x_cl, groups, nsample, mlp_module, idx=None, counts=None, total=None):
    """x_cl (P,3) float32 cuda grouped xyz (P = groups*nsample rows); mlp_module: the
    reference-shaped SharedMLP; idx: the ball-query indices (B,M,S) the rows came from.
    Padded copies inside a group are computed once (``compact_groups``).  The distinct-row count has to be on
    the host for that: a prefetched preparation brings it along (``counts`` / ``total``); otherwise the call
    waits for it -- one synchronisation at the very start of the step, where the host is far ahead of the
    device anyway (the unchanged caller's step: 407 vs 376 scenes/s).  ``CODA_SA_DEDUP=nosync`` only
    de-duplicates with a prefetched count, ``CODA_SA_DEDUP=0`` never.
    -> (groups, C_last)."""
    if idx is not None and mfma_eligible(mlp_module, nsample):
        packed = pack_groups(idx, x_cl, mfma_widths(mlp_module), dedup=os.environ.get("CODA_SA_DEDUP", "auto") != "0")
        return mfma_mlp_pool(packed, groups, nsample, mlp_module)
    layers = list(mlp_module.children())
    bns = [layer.bn.bn for layer in layers]
    params = []
    for layer in layers:
        params += [layer.conv.weight, layer.bn.bn.weight, layer.bn.bn.bias]
    training = bns[0].training
    dedup = None
    mode = os.environ.get("CODA_SA_DEDUP", "auto")
    if total is None and mode == "auto" and torch.cuda.is_current_stream_capturing():
        mode = "nosync"  # the row count cannot be read back inside a stream capture
    if idx is not None and mode != "0" and (total is not None or mode != "nosync"):
        compact = compact_groups(idx, x_cl, counts, total)
        if compact is not None:
            x_cl, roww, goff = compact
            dedup = (roww, goff)
    return _FusedMlpPool.apply(x_cl, groups, nsample, bns, training, dedup, *params)


def eligible(mlp_module, features, use_xyz, pooling, xyz):
    """The fused path covers the pre-encoder of the model: xyz-only input, max pooling,
    >= 2 layers of bias-free 1x1 Conv2d + BatchNorm2d + ReLU, fp32 on the GPU."""
    if features is not None or not use_xyz or pooling != "max" or not xyz.is_cuda or xyz.dtype != torch.float32:
        return False
    if os.environ.get("CODA_SA_MLP", "fused") == "layers":  # A/B switch: per-layer library ops
        return False
    layers = list(mlp_module.children())
    if len(layers) < 2:
        return False
    for i, layer in enumerate(layers):
        names = [n for n, _ in layer.named_children()]
        if names != ["conv", "bn", "activation"]:
            return False
        conv, bn = layer.conv, layer.bn.bn
        if conv.bias is not None or conv.kernel_size != (1, 1) or not isinstance(layer.activation, torch.nn.ReLU):
            return False
        if not isinstance(bn, (torch.nn.BatchNorm2d, torch.nn.SyncBatchNorm)) or not bn.affine:
            return False
        c = conv.out_channels
        if c % 4 or 1024 % c or (i == 0 and conv.in_channels != 3):
            return False
        if not bn.training and bn.running_mean is None:
            return False
    return True
