"""Building blocks of the fused pre-norm transformer layers (transformer.py).

Three autograd functions over the kernels of ``csrc/token_ln.hip`` / ``csrc/attention.hip``
and plain library GEMMs; they read the parameters of the reference-shaped modules
(``nn.LayerNorm``, ``MultiheadAttention.in_proj_weight`` ..., ``nn.Linear``) and change
neither their names nor their values:

* ``add_ln``   v = dropout(x + bias); s = res + v; y = LayerNorm(s); yp = y + pos -- the
  bias / Dropout / residual / LayerNorm / ``with_pos_embed`` chain between two GEMMs of
  models/transformer.py:457-494, 558-594 in one pass each way;
* ``mha``      in-projection GEMMs + fused attention core + out-projection GEMM as ONE node:
  the packed ``in_proj_weight`` gets its gradient written in place, slice by slice, instead
  of through three sliced views, zero-filled buffers and adds;
* ``ffn_act``  dropout(relu(h + bias)).
"""
import torch

from . import _lib, gemm
from . import attention_core as _core
from .linear_fn import _CHUNK, _MIN_ROWS


_FWD_X3 = __import__("os").environ.get("CODA_ATTN_FWD_X3", "1") != "0"


def _p(t):
    return t.data_ptr() if t is not None else None


def _stream():
    return _lib.current_stream_handle()


def _call(name, *args):
    lib = _lib.load()
    _lib.check(getattr(lib, name)(*args, _stream()), name)


def _colsum_into(out, x3, defer=None):
    """out (G*C) <- column sums of x3 (G, rows, C).  With a collector (gemm.DeferredWeightGrads) only the
    per-block partials are produced now; the reduction joins the collector's grouped launch."""
    g, rows, c = x3.shape
    if c % 4 or c > 1024 or not x3.is_contiguous():  # (the padded key / value gradient buffers come here as views)
        torch.sum(x3, 1, out=out.view(g, c))
        return
    blocks = _lib.load().coda_tok_colsum_blocks(rows, c)
    partials = torch.empty((g, blocks, c), dtype=torch.float32, device=x3.device)
    if defer is not None and rows > 0 and gemm.GROUPED_TN:
        _call("coda_tok_colsum_f32", _p(x3), g, rows, c, _p(partials), None)
        defer.add_colsum(partials, out, blocks, c, g)
        return
    _call("coda_tok_colsum_f32", _p(x3), g, rows, c, _p(partials), _p(out))


def _tn_into(out, dy, x):
    """out (Co,Ci) <- dy^T x with the row reduction split into chunks for long inputs."""
    p = dy.shape[0]
    part = gemm.x3_tn_partials(dy, x) if p >= _MIN_ROWS else None
    if part is not None:
        torch.sum(part, 0, out=out)
    elif p >= _MIN_ROWS and p % _CHUNK == 0:
        nc = p // _CHUNK
        part = torch.bmm(dy.view(nc, _CHUNK, -1).transpose(1, 2), x.view(nc, _CHUNK, -1))
        torch.sum(part, 0, out=out)
    else:
        gemm.mm_tn(dy, x, out=out)


class _AddLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, res, pos, gamma, beta, eps, p):
        rows, c = x.numel() // x.shape[-1], x.shape[-1]
        x = x.contiguous()
        res = res.contiguous() if res is not None else None
        pos = pos.contiguous() if pos is not None else None
        changes = bias is not None or res is not None or p > 0.0
        s = torch.empty_like(x) if changes else None
        y = torch.empty_like(x) if gamma is not None else None
        yp = torch.empty_like(x) if pos is not None else None
        mean = rstd = None
        if gamma is not None:
            mean = torch.empty(rows, dtype=torch.float32, device=x.device)
            rstd = torch.empty_like(mean)
        seed, seed_dev = _core._next_seed() if p > 0.0 else (0, None)
        _call("coda_tok_add_ln_fwd_f32", _p(x), _p(bias), _p(res), _p(pos), _p(gamma), _p(beta), rows, c, float(eps),
              float(p), seed, _p(seed_dev), _p(s), _p(y), _p(yp), _p(mean), _p(rstd))
        ctx.meta = (rows, c, float(p), seed, seed_dev, changes, bias is not None, res is not None, pos is not None)
        ctx.save_for_backward(s if changes else x, mean, rstd, gamma)
        ctx.set_materialize_grads(False)
        return s, y, yp

    @staticmethod
    def backward(ctx, ds, dy, dyp):
        rows, c, p, seed, seed_dev, changes, has_bias, has_res, has_pos = ctx.meta
        s, mean, rstd, gamma = ctx.saved_tensors
        if dy is None and dyp is None:
            gamma = None  # the normalised outputs were not used: only the residual stream flows
            if ds is None:
                return (None,) * 8
        ds = ds.contiguous() if ds is not None else None
        dy = dy.contiguous() if dy is not None else None
        dyp = dyp.contiguous() if dyp is not None else None
        blocks = _lib.load().coda_tok_add_ln_bwd_blocks(rows, c)
        partials = torch.empty((blocks, 3, c), dtype=torch.float32, device=s.device)
        dres = torch.empty_like(s) if (has_res or p == 0.0) else None
        dx = torch.empty_like(s) if p > 0.0 else None
        sums = torch.empty(3 * c, dtype=torch.float32, device=s.device)
        defer = getattr(ctx, "defer", None) if (rows > 0 and gemm.GROUPED_TN) else None
        _call("coda_tok_add_ln_bwd_f32", _p(dy), _p(dyp), _p(ds), _p(s), _p(mean), _p(rstd), _p(gamma), rows, c, p,
              seed, _p(seed_dev), _p(dres), _p(dx), _p(partials), None if defer is not None else _p(sums))
        if defer is not None:  # dgamma / dbeta / dbias are closed by the stack's grouped reduction
            defer.add_colsum(partials, sums, blocks, 3 * c)
        if dx is None:
            dx = dres
        has_ln = gamma is not None
        return (dx, sums[2 * c:] if has_bias else None, dres if has_res else None, dyp if has_pos else None,
                sums[:c] if has_ln else None, sums[c:2 * c] if has_ln else None, None, None)


def add_ln(x, norm=None, bias=None, res=None, pos=None, p=0.0):
    """-> (s, y, yp): s = res + dropout_p(x + bias) (``x`` itself when nothing is added),
    y = norm(s) (None without ``norm``), yp = y + pos (None without ``pos``)."""
    gamma = beta = None
    eps = 0.0
    if norm is not None:
        gamma, beta, eps = norm.weight, norm.bias, norm.eps
    s, y, yp = _AddLN.apply(x, bias, res, pos, gamma, beta, eps, float(p))
    return (x if s is None else s), y, yp


class _FfnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, bias, p):
        rows, c = h.numel() // h.shape[-1], h.shape[-1]
        h = h.contiguous()
        seed, seed_dev = _core._next_seed() if p > 0.0 else (0, None)
        a = torch.empty_like(h)
        _call("coda_tok_bias_relu_dropout_fwd_f32", _p(h), _p(bias), rows, c, float(p), seed, _p(seed_dev), _p(a))
        ctx.meta = (rows, c, float(p), bias is not None)
        ctx.save_for_backward(a)
        return a

    @staticmethod
    def backward(ctx, da):
        rows, c, p, has_bias = ctx.meta
        (a,) = ctx.saved_tensors
        da = da.contiguous()
        blocks = _lib.load().coda_tok_bias_relu_dropout_bwd_blocks(rows, c)
        partials = torch.empty((blocks, c), dtype=torch.float32, device=a.device)
        dz = torch.empty_like(a)
        db = torch.empty(c, dtype=torch.float32, device=a.device) if has_bias else None
        defer = getattr(ctx, "defer", None) if (has_bias and rows > 0 and gemm.GROUPED_TN) else None
        _call("coda_tok_bias_relu_dropout_bwd_f32", _p(da), _p(a), rows, c, p, _p(dz), _p(partials),
              None if defer is not None else _p(db))
        if defer is not None:
            defer.add_colsum(partials, db, blocks, c)
        return dz, db, None


def ffn_act(h, bias, p):
    return _FfnAct.apply(h, bias, float(p))


class _MHA(torch.autograd.Function):
    """xq (L,B,E), xk / xv (S,B,E) -> attention output projected by w_out, WITHOUT the
    out-projection bias (the caller's add_ln adds it): (L,B,E)."""

    @staticmethod
    def forward(ctx, xq, xk, xv, w_in, b_in, w_out, mask_u8, nheads, p):
        lib = _lib.load()
        tgt_len, bsz, e = xq.shape
        src_len = xk.shape[0]
        d = e // nheads
        same_qk = xk is xq
        same_kv = xv is xk
        xq2 = xq.reshape(-1, e)
        xk2 = xq2 if same_qk else xk.reshape(-1, e)
        xv2 = xk2 if same_kv else xv.reshape(-1, e)
        if same_qk and same_kv:
            qkv = gemm.linear(xq2, w_in, b_in)
            q, k, v = qkv[:, :e], qkv[:, e:2 * e], qkv[:, 2 * e:]
            ldq = ldk = ldv = 3 * e
        elif same_qk:
            qk = gemm.linear(xq2, w_in[:2 * e], b_in[:2 * e])
            q, k = qk[:, :e], qk[:, e:]
            v = gemm.linear(xv2, w_in[2 * e:], b_in[2 * e:])
            ldq = ldk = 2 * e
            ldv = e
        else:
            q = gemm.linear(xq2, w_in[:e], b_in[:e])
            k = gemm.linear(xk2, w_in[e:2 * e], b_in[e:2 * e])
            v = gemm.linear(xv2, w_in[2 * e:], b_in[2 * e:])
            ldq = ldk = ldv = e
        attn = torch.empty((tgt_len * bsz, e), dtype=torch.float32, device=xq.device)
        lse = torch.empty((bsz, nheads, tgt_len), dtype=torch.float32, device=xq.device)
        seed, seed_dev = _core._next_seed() if p > 0.0 else (0, None)
        scale = 1.0 / (d ** 0.5)
        dt = _lib.opt("mfma_dtype")  # per-call MFMA operand type (this thread's option), kept for the backward
        # Long unmasked sequences in fp32 mode (the encoder's 2048 x 2048): the FORWARD core runs its products as three
        # bf16 pieces on the real matrix cores (mode 2 of include/coda_attention.h: fp32-level accuracy,
        # tests/test_attention_x3_gpu.py; 258 against 352 us per layer, tools/bench_attn.py).  The backward is NOT run
        # in that mode as a whole (594 against 651 us: the probabilities would be split per element); since round 6 its
        # own kernels put the products whose operands are staged tiles -- S, dP, dQ = dS K -- on the bf16 matrix cores
        # and keep dV / dK on the fp32 MFMA (csrc/attention.hip, mha_bwd_dkv_x3_kernel / mha_bwd_dq_x3_kernel: 567 + 150
        # against 644 + 185 us), with alternating signs against the bf16 accumulate's drift (csrc/gemm_x3.hip).
        # CODA_ATTN_FWD_X3=0: fp32 MFMA forward too (A/B).
        fwd_dt = dt
        if (_FWD_X3 and mask_u8 is None and tgt_len >= 1024 and src_len >= 1024 and d == 64
                and (dt == 0 or (dt < 0 and lib.coda_mha_get_mfma_dtype() == 0))):
            fwd_dt = 2
        _lib.check(lib.coda_mha_fwd_opt_f32(_p(q), _p(k), _p(v), _p(mask_u8), _p(attn), _p(lse), bsz, nheads, tgt_len,
                                            src_len, d, ldq, ldk, ldv, scale, float(p), seed, _p(seed_dev), fwd_dt,
                                            _stream()),
                   "mha_fwd")
        out = gemm.linear(attn, w_out).view(tgt_len, bsz, e)
        ctx.meta = (tgt_len, src_len, bsz, e, nheads, ldq, ldk, ldv, scale, float(p), seed, seed_dev, same_qk, same_kv)
        ctx.mfma_dtype = dt
        ctx.save_for_backward(xq2, xk2, xv2, q, k, v, attn, lse, w_in, w_out, mask_u8)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        tgt_len, src_len, bsz, e, nheads, ldq, ldk, ldv, scale, p, seed, seed_dev, same_qk, same_kv = ctx.meta
        xq2, xk2, xv2, q, k, v, attn, lse, w_in, w_out, mask_u8 = ctx.saved_tensors
        dev = dout.device
        d = e // nheads
        dout2 = dout.reshape(-1, e)
        # a stack-level backward (fused_blocks._DecoderStack) hands in a collector: weight gradients are issued
        # together at its end (gemm.DeferredWeightGrads) instead of one launch-sized GEMM each
        defer = getattr(ctx, "defer", None)
        tn_into = defer.add if defer is not None else _tn_into
        dw_out = torch.empty_like(w_out)
        tn_into(dw_out, dout2.contiguous(), attn)
        dattn = gemm.mm(dout2, w_out)
        rq, rk = tgt_len * bsz, src_len * bsz
        if rq == rk:
            dqkv = torch.empty((3, rq, e), dtype=torch.float32, device=dev)
            dq, dk, dv = dqkv[0], dqkv[1], dqkv[2]
        else:
            dqkv = None
            dq = torch.empty((rq, e), dtype=torch.float32, device=dev)
            dkv = torch.empty((2, rk, e), dtype=torch.float32, device=dev)
            dk, dv = dkv[0], dkv[1]
        delta = torch.empty((bsz, nheads, tgt_len), dtype=torch.float32, device=dev)
        # long unmasked sequences (the encoder): dS through a workspace, dQ as one GEMM (include/coda_attention.h)
        ws, ws_bytes = (_core.backward_workspace(bsz, nheads, tgt_len, src_len, d, dev, ctx.mfma_dtype)
                        if mask_u8 is None else (None, 0))
        _lib.check(lib.coda_mha_bwd_ws_f32(_p(q), _p(k), _p(v), _p(mask_u8), _p(attn), _p(lse), _p(dattn), _p(dq),
                                           _p(dk), _p(dv), _p(delta), bsz, nheads, tgt_len, src_len, d, ldq, ldk, ldv,
                                           0, 0, 0, scale, p, seed, _p(seed_dev), _p(ws), ws_bytes, ctx.mfma_dtype,
                                           _stream()), "mha_bwd")
        dw_in = torch.empty_like(w_in)
        db_in = torch.empty(3 * e, dtype=torch.float32, device=dev)
        if dqkv is not None:
            _colsum_into(db_in, dqkv, defer)
        else:
            _colsum_into(db_in[:e], dq.unsqueeze(0), defer)
            _colsum_into(db_in[e:], dkv, defer)
        tn_into(dw_in[:e], dq, xq2)
        tn_into(dw_in[e:2 * e], dk, xk2)
        tn_into(dw_in[2 * e:], dv, xv2)
        need_q, need_k, need_v = ctx.needs_input_grad[:3]
        dxq = dxk = dxv = None
        if same_qk and same_kv:
            if need_q:
                dxq = gemm.mm(dq, w_in[:e])
                gemm.mm(dk, w_in[e:2 * e], out=dxq, accumulate=True)
                gemm.mm(dv, w_in[2 * e:], out=dxq, accumulate=True)
        elif same_qk:
            if need_q:
                dxq = gemm.mm(dq, w_in[:e])
                gemm.mm(dk, w_in[e:2 * e], out=dxq, accumulate=True)
            if need_v:
                dxv = gemm.mm(dv, w_in[2 * e:])
        else:
            if need_q:
                dxq = gemm.mm(dq, w_in[:e])
            if need_k:
                dxk = gemm.mm(dk, w_in[e:2 * e])
            if need_v:
                if same_kv:
                    dxk = gemm.mm(dv, w_in[2 * e:]) if dxk is None else gemm.mm(dv, w_in[2 * e:], out=dxk, accumulate=True)
                else:
                    dxv = gemm.mm(dv, w_in[2 * e:])

        def shaped(t, n):
            return t.view(n, bsz, e) if t is not None else None

        return (shaped(dxq, tgt_len), shaped(dxk, src_len), shaped(dxv, src_len), dw_in, db_in, dw_out, None, None,
                None)


def mha(attn_module, xq, xk, xv, mask_u8=None):
    """Attention block with the parameters of a ``MultiheadAttention`` module; the output
    lacks ``out_proj.bias`` (pass it as ``bias`` to the following ``add_ln``)."""
    p = attn_module.dropout if attn_module.training else 0.0
    return _MHA.apply(xq, xk, xv, attn_module.in_proj_weight, attn_module.in_proj_bias, attn_module.out_proj.weight,
                      mask_u8, attn_module.num_heads, float(p))
