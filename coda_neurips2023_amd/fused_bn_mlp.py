"""GenericMLP stacks on channels-last tokens: (batched) library GEMMs + the streaming
batch-norm kernels of ``csrc/token_bn.hip`` (``include/coda_token_ops.h``).

The reference evaluates its six prediction heads one after the other as
Conv1d(k=1) -> BatchNorm1d -> ReLU -> Dropout stacks on ``(layers*batch, C, queries)``
(models/model_3detr.py:1617-1660, models/helpers.py:45-112), and the encoder->decoder
projection the same way on ``(batch, C, points)`` (:1866-1868).  A 1x1 convolution over
tokens is a GEMM on the ``(tokens, C)`` matrix the transformer already holds, and G heads
over the same input are one batched GEMM, so ``run_stacks`` computes all heads at once on
``(G, tokens, C)`` activations with one statistics pass and one normalise/activate/dropout
pass between two GEMMs.  Parameters stay in the reference-shaped modules
(``layers.{i}.weight`` ...); train-mode batch statistics, running-statistics updates and
SyncBatchNorm reductions are those of ``torch.nn.BatchNorm1d`` / ``SyncBatchNorm``.
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib, gemm
from . import attention_core as _core

_SPLIT_ROWS = 2048
_NARROW = 16  # tails with at most this many outputs are batched (zero-padded) into one GEMM


def _p(t):
    return t.data_ptr() if t is not None else None


def _call(name, *args):
    lib = _lib.load()
    _lib.check(getattr(lib, name)(*args, _lib.current_stream_handle()), name)


def _parts(groups, rows, c):
    """Partial sums per group and channel the statistics kernels leave for this shape (include/coda_token_ops.h)."""
    n = _lib.load().coda_tok_bn_parts(groups, rows, c)
    if n <= 0:
        raise RuntimeError("coda_tok_bn_parts(%d, %d, %d) failed: %d" % (groups, rows, c, n))
    return n


def _is_sync(bn):
    return isinstance(bn, nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized() \
        and dist.get_world_size(bn.process_group) > 1


_PAD_ROWS = {}


def _pad_rows(outs, width, device):
    """Index of row r of head g in the zero-padded (len(outs)*width) stack, cached per shape."""
    key = (outs, width, str(device))
    idx = _PAD_ROWS.get(key)
    if idx is None:
        idx = torch.tensor([g * width + r for g, o in enumerate(outs) for r in range(o)], dtype=torch.int64,
                           device=device)
        _PAD_ROWS[key] = idx
    return idx


def parse(mlp):
    """GenericMLP -> ([(dense, bn, relu, dropout_p), ...], tail dense or None), or None when
    the stack has a shape the fused path does not cover."""
    mods = list(mlp.layers.children())
    blocks, i = [], 0
    tail = None
    while i < len(mods):
        dense = mods[i]
        if isinstance(dense, nn.Conv1d):
            if dense.kernel_size != (1,) or dense.stride != (1,) or dense.groups != 1 or dense.padding != (0,):
                return None
        elif not isinstance(dense, nn.Linear):
            return None
        i += 1
        if i == len(mods):
            tail = dense
            break
        bn = mods[i]
        if not isinstance(bn, (nn.BatchNorm1d, nn.SyncBatchNorm)) or dense.bias is not None:
            return None
        if not bn.affine or not bn.track_running_stats or bn.momentum is None:
            return None
        i += 1
        relu = i < len(mods) and isinstance(mods[i], nn.ReLU)
        i += int(relu)
        p = 0.0
        if i < len(mods) and isinstance(mods[i], nn.Dropout):
            p = float(mods[i].p)
            i += 1
        cout = dense.weight.shape[0]
        if cout % 4 or cout > 1024 or 256 % (cout // 4) or not 0.0 <= p < 1.0:
            return None
        blocks.append((dense, bn, relu, p))
    if not blocks:
        return None
    return blocks, tail


def eligible(mlps, x):
    """All stacks have the same block structure, are in training mode, fp32 on the GPU."""
    if not x.is_cuda or x.dtype != torch.float32:
        return None
    parsed = [parse(m) for m in mlps]
    if any(p is None for p in parsed):
        return None
    first = parsed[0][0]
    for blocks, _ in parsed:
        if len(blocks) != len(first):
            return None
        for (d, bn, relu, p), (d0, bn0, relu0, p0) in zip(blocks, first):
            if d.weight.shape[:2] != d0.weight.shape[:2] or relu != relu0 or p != p0 or bn.eps != bn0.eps \
                    or bn.momentum != bn0.momentum or not bn.training or type(bn) is not type(bn0):
                return None
            if _is_sync(bn) and bn.process_group is not bn0.process_group:
                return None
    return parsed


def _split_k_tn(dz, a, defer=None, out=None):
    """dz (G,T,Co), a (G,T,Ci) contiguous -> dz^T a (G,Co,Ci) with the T reduction split.  With a collector
    (gemm.DeferredWeightGrads) the sum over the row chunks is closed by the collector's grouped launch: the returned
    tensor (``out`` if given: (G,Co,Ci), contiguous) holds its values only after ``defer.flush()``."""
    g, t, co = dz.shape
    ci = a.shape[-1]
    if defer is not None and t >= 2 * _SPLIT_ROWS:
        # the x3 kernel's partial sums per token slice (gemm.x3_tn_partials), one launch per group
        first = gemm.x3_tn_partials(dz[0], a[0])
        if first is not None:
            part = first if g == 1 else torch.empty((g,) + tuple(first.shape), dtype=torch.float32, device=dz.device)
            if g > 1:
                part[0].copy_(first)
                for k in range(1, g):
                    gemm.x3_tn_partials(dz[k], a[k], out=part[k])
            if out is None:
                out = torch.empty((g, co, ci), dtype=torch.float32, device=dz.device)
            defer.add_colsum(part, out, first.shape[0], co * ci, g)
            return out
    if t >= 2 * _SPLIT_ROWS and t % _SPLIT_ROWS == 0:
        nc = t // _SPLIT_ROWS
        part = torch.bmm(dz.view(g * nc, _SPLIT_ROWS, co).transpose(1, 2), a.view(g * nc, _SPLIT_ROWS, ci))
        if defer is not None:
            if out is None:
                out = torch.empty((g, co, ci), dtype=torch.float32, device=dz.device)
            defer.add_colsum(part, out, nc, co * ci, g)   # partials (G, nc, Co*Ci) -> out (G, Co*Ci)
            return out
        res = part.view(g, nc, co, ci).sum(1)
    else:
        res = torch.bmm(dz.transpose(1, 2), a)
    if out is not None:
        out.copy_(res)
        return out
    return res


class _HiddenStack(torch.autograd.Function):
    """x (T, Cin) shared by G stacks -> (G, T, C_last) after every (dense, BN, ReLU, dropout)
    block, or, with tail layers, the G outputs (T, out_g) of the final plain dense layers.
    params: per block, per group: dense weight, bn weight, bn bias; then per group the tail
    weight and bias (bias may be None)."""

    @staticmethod
    def forward(ctx, x, meta, *params):
        groups, blocks, has_tail = meta  # blocks: list of (bns [G], relu, p)
        nb = len(blocks)
        dev = x.device
        t = x.shape[0]
        pit = iter(params)
        ws, gammas, betas = [], [], []
        for _ in range(nb):
            w, gm, bt = [], [], []
            for _ in range(groups):
                w.append(next(pit).flatten(1))
                gm.append(next(pit))
                bt.append(next(pit))
            ws.append(torch.stack(w) if groups > 1 else w[0].unsqueeze(0))
            gammas.append(torch.stack(gm) if groups > 1 else gm[0].unsqueeze(0))
            betas.append(torch.stack(bt) if groups > 1 else bt[0].unsqueeze(0))

        zs, acts, prms, seeds = [], [], [], []
        cur = None  # (G,T,C) input of the block (None: x)
        new_stats, run_stats, counters = [], [], []
        for i, (bns, relu, p) in enumerate(blocks):
            c = ws[i].shape[1]
            if cur is None:
                if groups == 1:
                    z = gemm.linear(x, ws[i][0]).unsqueeze(0)
                else:  # one batched GEMM over the heads; x is broadcast with batch stride 0
                    z = torch.bmm(x.unsqueeze(0).expand(groups, -1, -1), ws[i].transpose(1, 2))
            elif groups == 1:
                z = gemm.linear(cur[0], ws[i][0]).unsqueeze(0)
            else:
                z = torch.bmm(cur, ws[i].transpose(1, 2))
            # one partial sum per row block; the finalize kernel adds them (no atomics, no memset)
            nparts = _parts(groups, t, c)
            sums = torch.empty((nparts, groups, 2, c), dtype=torch.float64, device=dev)
            _call("coda_tok_bn_stats_f32", _p(z), groups, t, c, _p(sums))
            world = 1
            if _is_sync(bns[0]):
                world = dist.get_world_size(bns[0].process_group)
                sums, nparts = sums.sum(0), 1
                dist.all_reduce(sums, group=bns[0].process_group)
            n = float(t * world)
            prm = torch.empty((groups, 4, c), dtype=torch.float32, device=dev)
            stat = torch.empty((groups, 2, c), dtype=torch.float32, device=dev)
            _call("coda_tok_bn_finalize_f32", _p(sums), nparts, _p(gammas[i]), _p(betas[i]), groups, c, n,
                  float(bns[0].eps), _p(prm), _p(stat))
            for g, bn in enumerate(bns):
                run_stats += [bn.running_mean, bn.running_var]
                new_stats += [stat[g, 0], stat[g, 1]]
                counters.append(bn.num_batches_tracked)
            seed, seed_dev = _core._next_seed() if p > 0.0 else (0, None)
            act = torch.empty_like(z)
            _call("coda_tok_bn_act_f32", _p(z), _p(prm), groups, t, c, int(relu), p, seed, _p(seed_dev), _p(act))
            zs.append(z)
            acts.append(act)
            prms.append(prm)
            seeds.append((seed, seed_dev, n))
            cur = act
        with torch.no_grad():  # running statistics of all BN modules in two multi-tensor launches
            torch._foreach_lerp_(run_stats, new_stats, float(blocks[0][0][0].momentum))
            torch._foreach_add_(counters, 1)

        ctx.meta = meta
        ctx.seeds = seeds
        ctx.wshapes = [params[3 * groups * i].shape for i in range(nb)]
        tails = [next(pit) for _ in range(2 * groups)] if has_tail else []
        ctx.tail_shapes = [w.shape for w in tails[0::2]]
        ctx.tail_bias = [b is not None for b in tails[1::2]]
        tail_ws = [w.flatten(1) for w in tails[0::2]]
        # Narrow tails (box heads: 2..12 outputs) are GEMMs with a tiny N (forward) / K (backward)
        # that the library runs at a few % of its rate: the leading run of narrow heads is
        # evaluated as ONE batched GEMM on weights zero-padded to a common width.
        ns = 0
        while has_tail and ns < groups and tail_ws[ns].shape[0] <= _NARROW and tails[2 * ns + 1] is not None:
            ns += 1
        ns = ns if ns >= 2 else 0
        wpad = None
        if ns:
            outs_n = tuple(int(tail_ws[g].shape[0]) for g in range(ns))
            width = -(-max(outs_n) // 4) * 4
            rows = _pad_rows(outs_n, width, dev)                      # row of the padded stack for every real row
            c_last = tail_ws[0].shape[1]
            wpad = torch.zeros((ns * width, c_last), dtype=torch.float32, device=dev)
            wpad.index_copy_(0, rows, torch.cat([tail_ws[g] for g in range(ns)]))
            wpad = wpad.view(ns, width, c_last)
            bpad = torch.zeros(ns * width, dtype=torch.float32, device=dev)
            bpad.index_copy_(0, rows, torch.cat([tails[2 * g + 1] for g in range(ns)]))
            bpad = bpad.view(ns, width)
        ctx.narrow = ns
        ctx.save_for_backward(x, *zs, *acts, *prms, *ws, *gammas, *tail_ws, *([wpad] if ns else []))
        if not has_tail:
            return acts[-1]
        outs = []
        if ns:
            small = torch.baddbmm(bpad.unsqueeze(1), acts[-1][:ns], wpad.transpose(1, 2))  # (ns, T, width)
            outs = [small[g, :, :tail_ws[g].shape[0]] for g in range(ns)]
        outs += [torch.addmm(tails[2 * g + 1], acts[-1][g], tail_ws[g].t()) if tails[2 * g + 1] is not None
                 else torch.mm(acts[-1][g], tail_ws[g].t()) for g in range(ns, groups)]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        groups, blocks, has_tail = ctx.meta
        nb = len(blocks)
        saved = ctx.saved_tensors
        x = saved[0]
        zs = saved[1:1 + nb]
        acts = saved[1 + nb:1 + 2 * nb]
        prms = saved[1 + 2 * nb:1 + 3 * nb]
        ws = saved[1 + 3 * nb:1 + 4 * nb]
        gammas = saved[1 + 4 * nb:1 + 5 * nb]
        ns = ctx.narrow
        tail_ws = saved[1 + 5 * nb:1 + 5 * nb + groups] if has_tail else ()
        dev = x.device
        t = x.shape[0]
        grads = [None] * (3 * groups * nb)
        tail_grads = []
        # the sums over the row chunks of the ~12 weight-gradient products close in one grouped launch at the end
        defer = gemm.DeferredWeightGrads(sums_only=True) if (gemm.DEFER_SUMS and x.is_cuda) else None
        if has_tail:
            # gradients of the G tail layers; dA is assembled in place, group by group
            dout = None
            da = torch.empty_like(acts[-1])
            if ns:  # the narrow heads as one batched problem on zero-padded gradients
                wpad = saved[-1]
                width = wpad.shape[1]
                dpad = torch.zeros((ns, t, width), dtype=torch.float32, device=dev)
                for g in range(ns):
                    dpad[g, :, :tail_ws[g].shape[0]].copy_(douts[g])
                torch.bmm(dpad, wpad, out=da[:ns])
                dwp = _split_k_tn(dpad, acts[-1][:ns], defer)                              # (ns, width, C)
                dbp = dpad.sum(1)
                for g in range(ns):
                    og = tail_ws[g].shape[0]
                    tail_grads += [dwp[g, :og].reshape(ctx.tail_shapes[g]), dbp[g, :og]]
            for g in range(ns, groups):
                dg = douts[g].contiguous()
                torch.mm(dg, tail_ws[g], out=da[g])
                dw = _split_k_tn(dg.unsqueeze(0), acts[-1][g].unsqueeze(0), defer)[0]
                tail_grads += [dw.view(ctx.tail_shapes[g]), _colsum(dg) if ctx.tail_bias[g] else None]
        else:
            dout = douts[0]
            da = dout.contiguous()
        dx = None
        for i in range(nb - 1, -1, -1):
            bns, relu, p = blocks[i]
            seed, seed_dev, n = ctx.seeds[i]
            c = ws[i].shape[1]
            nparts = _parts(groups, t, c)
            sums = torch.empty((nparts, groups, 2, c), dtype=torch.float64, device=dev)
            _call("coda_tok_bn_act_bwd_stats_f32", _p(da), _p(zs[i]), _p(prms[i]), groups, t, c, int(relu), p, seed,
                  _p(seed_dev), _p(sums))
            total = None  # single process: the local sums
            if _is_sync(bns[0]):
                sums, nparts = sums.sum(0), 1
                total = sums.clone()
                dist.all_reduce(total, group=bns[0].process_group)
            prmb = torch.empty((groups, 3, c), dtype=torch.float32, device=dev)
            dgamma = torch.empty((groups, c), dtype=torch.float32, device=dev)
            dbeta = torch.empty((groups, c), dtype=torch.float32, device=dev)
            _call("coda_tok_bn_bwd_finalize_f32", _p(sums), nparts, _p(total), _p(gammas[i]), _p(prms[i]), groups, c, n,
                  _p(prmb), _p(dgamma), _p(dbeta))
            dz = da if (dout is None or da.data_ptr() != dout.data_ptr()) else torch.empty_like(da)
            _call("coda_tok_bn_act_bwd_apply_f32", _p(da), _p(zs[i]), _p(prms[i]), _p(prmb), groups, t, c, int(relu),
                  p, seed, _p(seed_dev), _p(dz))
            # weight gradients (split-K) and the gradient of the block input
            if i > 0:
                dw = _split_k_tn(dz, acts[i - 1], defer)
                da = gemm.mm(dz[0], ws[i][0]).unsqueeze(0) if groups == 1 else torch.bmm(dz, ws[i])
            else:
                xs = x.unsqueeze(0)
                if groups == 1:
                    dw = _split_k_tn(dz, xs, defer)
                    dx = gemm.mm(dz[0], ws[0][0]) if ctx.needs_input_grad[0] else None
                else:
                    dw = torch.empty((groups, dz.shape[2], xs.shape[2]), dtype=torch.float32, device=dev)
                    for g in range(groups):  # (x is shared by the groups: one product per group, written in place)
                        _split_k_tn(dz[g:g + 1], xs, defer, out=dw[g:g + 1])
                    if ctx.needs_input_grad[0]:
                        dx = torch.mm(dz[0], ws[0][0])
                        for g in range(1, groups):
                            dx.addmm_(dz[g], ws[0][g])
            for g in range(groups):
                base = 3 * (groups * i + g)
                grads[base] = dw[g].view(ctx.wshapes[i])
                grads[base + 1] = dgamma[g]
                grads[base + 2] = dbeta[g]
        if defer is not None:
            defer.flush()
        return (dx, None, *grads, *tail_grads)


def _colsum(x2):
    """(rows, C) -> (C,) column sums (bias gradient)."""
    rows, c = x2.shape
    if c % 4 or c > 1024 or 256 % (c // 4):
        return x2.sum(0)
    blocks = _lib.load().coda_tok_colsum_blocks(rows, c)
    partials = torch.empty((blocks, c), dtype=torch.float32, device=x2.device)
    out = torch.empty(c, dtype=torch.float32, device=x2.device)
    _call("coda_tok_colsum_f32", _p(x2), 1, rows, c, _p(partials), _p(out))
    return out


def run_stacks(x, parsed):
    """x (T, Cin) float32 cuda, parsed = eligible(mlps, x).  Stacks with a final plain dense
    layer -> list of G outputs (T, out_g); stacks ending in a block -> (G, T, C_last)."""
    has_tail = all(tail is not None for _, tail in parsed)
    if not has_tail and any(tail is not None for _, tail in parsed):
        raise RuntimeError("stacks with and without a final dense layer cannot be mixed")
    out = _apply(x, parsed, has_tail)
    return list(out) if has_tail else out


def hidden_stack(x, parsed):
    """-> (G, T, C_last): the activations after the last (dense, BN, ReLU, dropout) block."""
    return _apply(x, parsed, False)


def _apply(x, parsed, has_tail):
    groups = len(parsed)
    nb = len(parsed[0][0])
    blocks = []
    params = []
    for i in range(nb):
        bns = [parsed[g][0][i][1] for g in range(groups)]
        blocks.append((bns, parsed[0][0][i][2], parsed[0][0][i][3]))
        for g in range(groups):
            dense, bn, _, _ = parsed[g][0][i]
            params += [dense.weight, bn.weight, bn.bias]
    if has_tail:
        for g in range(groups):
            params += [parsed[g][1].weight, parsed[g][1].bias]
    return _HiddenStack.apply(x.contiguous(), (groups, blocks, has_tail), *params)
