"""coda_gemm_f32 (include/coda_gemm.h) through gemm.py against torch.mm: plain, biased, accumulating,
and on row / column slices of packed buffers (the strides the attention blocks pass)."""
import pytest
import torch

from coda_neurips2023_amd import gemm

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("m,n,k", [(2048, 256, 256), (16384, 768, 256), (2048, 128, 256), (33, 7, 19), (1, 512, 256),
                                   (256, 256, 2048)])
def test_linear_mm_mm_tn_match_torch(dev, m, n, k):
    g = torch.Generator().manual_seed(m + n + k)
    x = torch.randn(m, k, generator=g).to(dev)
    w = torch.randn(n, k, generator=g).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    assert rel(gemm.linear(x, w), x @ w.t()) < 1e-5
    assert rel(gemm.linear(x, w, b), torch.addmm(b, x, w.t())) < 1e-5
    dy = torch.randn(m, n, generator=g).to(dev)
    assert rel(gemm.mm(dy, w), dy @ w) < 1e-5                    # input gradient
    assert rel(gemm.mm_tn(dy, x), dy.t() @ x) < 1e-5             # weight gradient
    # twice the same shape: the cached plan is reused
    assert rel(gemm.mm_tn(dy, x), dy.t() @ x) < 1e-5


def test_strided_operands_and_accumulation(dev):
    g = torch.Generator().manual_seed(0)
    e, rows = 256, 2048
    w_in = torch.randn(3 * e, e, generator=g).to(dev)            # packed in_proj weight
    b_in = torch.randn(3 * e, generator=g).to(dev)
    x = torch.randn(rows, e, generator=g).to(dev)
    qkv = gemm.linear(x, w_in, b_in)                             # (rows, 3e)
    k = qkv[:, e:2 * e]                                          # column slice: row stride 3e
    ref_k = torch.addmm(b_in[e:2 * e], x, w_in[e:2 * e].t())
    assert rel(k, ref_k) < 1e-5
    assert rel(gemm.linear(x, w_in[e:2 * e], b_in[e:2 * e]), ref_k) < 1e-5      # row slice of the weight
    assert rel(gemm.mm(k, w_in[e:2 * e]), ref_k @ w_in[e:2 * e]) < 1e-5        # strided A
    out = torch.zeros(3 * e, e, device=dev)
    gemm.mm_tn(k, x, out=out[e:2 * e])                            # into a row block of a packed gradient
    assert rel(out[e:2 * e], ref_k.t() @ x) < 1e-5 and float(out[:e].abs().max()) == 0.0
    acc = gemm.mm(k, w_in[e:2 * e])
    gemm.mm(qkv[:, :e], w_in[:e], out=acc, accumulate=True)
    assert rel(acc, ref_k @ w_in[e:2 * e] + qkv[:, :e] @ w_in[:e]) < 1e-5
    tn = gemm.mm_tn(k, x)
    gemm.mm_tn(qkv[:, :e], x, out=tn, accumulate=True)
    assert rel(tn, ref_k.t() @ x + qkv[:, :e].t() @ x) < 1e-5


def test_side_stream_gets_its_own_workspace(dev):
    x = torch.randn(4096, 256, device=dev)
    w = torch.randn(256, 256, device=dev)
    ref = x @ w.t()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = gemm.linear(x, w)
    torch.cuda.current_stream().wait_stream(s)
    assert rel(out, ref) < 1e-5


def test_invalid_arguments_are_reported(dev):
    from coda_neurips2023_amd import _lib
    lib = _lib.load()
    x = torch.randn(8, 8, device=dev)
    st = lib.coda_gemm_f32(0, 0, 8, 8, 8, x.data_ptr(), 4, x.data_ptr(), 8, x.data_ptr(), 8, None, 0,
                           _lib.current_stream_handle())
    assert st == _lib.CODA_EINVAL  # lda < k


@pytest.mark.parametrize("m,n,k,transb", [(2048, 256, 256, True), (2048, 256, 256, False), (64, 64, 128, True),
                                          (256, 128, 384, False), (16384, 256, 256, True), (16384, 128, 256, False),
                                          (4096, 192, 96, True), (128, 64, 32, False)])
def test_own_sgemm_kernels(dev, m, n, k, transb):
    """coda_sgemm_f32 (csrc/gemm_nn.hip): the split-K kernel for launch-sized problems and the 64x64-tile kernel,
    both operand forms, bias, accumulation, row-strided operands; against fp64."""
    from coda_neurips2023_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(m + n + k)
    a_full = torch.randn(m, k + 8, generator=g).to(dev)
    a = a_full[:, 4:4 + k]  # row stride k + 8, 16-B aligned offset
    b = (torch.randn(n, k, generator=g) if transb else torch.randn(k, n, generator=g)).to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    out = torch.full((m, n + 4), 3.0, device=dev)[:, :n]
    ref = a.double() @ (b.double().t() if transb else b.double())

    def call(bias_t, acc):
        st = lib.coda_sgemm_f32(1 if transb else 0, m, n, k, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                                out.data_ptr(), out.stride(0), bias_t.data_ptr() if bias_t is not None else None, acc,
                                _lib.current_stream_handle())
        assert st == 0, st
    call(bias, 0)
    assert rel(out, (ref + bias.double()).float()) < 1e-5
    call(None, 1)
    assert rel(out, (2 * ref + bias.double()).float()) < 1e-5
    # shapes outside the kernel's tiling are refused, not mangled
    st = lib.coda_sgemm_f32(1, 48, 64, 128, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(),
                            out.stride(0), None, 0, _lib.current_stream_handle())
    assert st == _lib.CODA_ENOSPC


@pytest.mark.parametrize("m,n,k,p", [(2048, 256, 256, 0.1), (2048, 256, 256, 0.0), (1024, 512, 512, 0.3), (64, 64, 128, 0.5)])
def test_sgemm_with_relu_dropout_epilogue_equals_the_two_passes(dev, m, n, k, p):
    """coda_sgemm_relu_dropout_f32 = coda_sgemm_f32 followed by coda_tok_bias_relu_dropout_fwd_f32 with the same seed,
    element for element (same sums, same keep decisions); shapes outside the launch-sized kernel are refused."""
    from coda_neurips2023_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).to(dev)
    w = torch.randn(n, k, generator=g).to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    seed = 0x1234567890ABCDEF
    fused = torch.empty(m, n, device=dev)
    st = lib.coda_sgemm_relu_dropout_f32(1, m, n, k, a.data_ptr(), k, w.data_ptr(), k, fused.data_ptr(), n, bias.data_ptr(),
                                         p, seed, _lib.current_stream_handle())
    assert st == 0, st
    two = torch.empty(m, n, device=dev)
    assert lib.coda_sgemm_f32(1, m, n, k, a.data_ptr(), k, w.data_ptr(), k, two.data_ptr(), n, None, 0,
                              _lib.current_stream_handle()) == 0
    assert lib.coda_tok_bias_relu_dropout_fwd_f32(two.data_ptr(), bias.data_ptr(), m, n, p, seed, None, two.data_ptr(),
                                                  _lib.current_stream_handle()) == 0
    assert torch.equal(fused, two)
    kept = float((fused != 0).float().mean())
    assert abs(kept - 0.5 * (1 - p)) < 0.05          # relu halves, dropout keeps 1 - p
    st = lib.coda_sgemm_relu_dropout_f32(1, 16384, 256, 256, a.data_ptr(), 256, w.data_ptr(), 256, fused.data_ptr(), 256,
                                         None, p, seed, _lib.current_stream_handle())   # (refused before any access)
    assert st == _lib.CODA_ENOSPC
    # the operand checks of coda_sgemm_f32 (round 3's advisor finding): k == 0, a row stride shorter than the row
    args = lambda kk, lda, ldb: lib.coda_sgemm_relu_dropout_f32(1, m, n, kk, a.data_ptr(), lda, w.data_ptr(), ldb,
                                                                fused.data_ptr(), n, None, p, seed,
                                                                _lib.current_stream_handle())
    assert args(0, k, k) == _lib.CODA_EINVAL
    assert args(k, k - 4, k) == _lib.CODA_EINVAL and args(k, k, k - 4) == _lib.CODA_EINVAL


def test_grouped_weight_gradients(dev):
    """coda_grouped_gemm_tn_f32 through gemm.DeferredWeightGrads: many dy^T x products in one launch -- mixed
    shapes, slices of packed buffers as operands and outputs, more problems than one launch carries, shapes the
    kernel does not take (computed on the spot), bit-identical results on a second run (fixed summation order)."""
    g = torch.Generator().manual_seed(3)
    shapes = [(2048, 256, 256)] * 70 + [(2048, 768, 256), (64, 64, 64), (8, 64, 128), (16384, 128, 256), (2048, 256, 128),
                                        (100, 64, 64), (2048, 96, 256)]   # the last two: rows % 8 / cols % 64 fail
    probs = []
    for rows, m, n in shapes:
        dy = torch.randn(rows, m, generator=g).to(dev)
        x = torch.randn(rows, n, generator=g).to(dev)
        probs.append((dy, x))
    packed_dy = torch.randn(2048, 3 * 256, generator=g).to(dev)   # column slices (row stride 768)
    packed_x = torch.randn(2048, 512, generator=g).to(dev)
    packed_out = torch.full((3, 256, 256), float("nan"), device=dev)
    outs = []
    for rep in range(2):
        d = gemm.DeferredWeightGrads()
        res = []
        for dy, x in probs:
            out = torch.full((dy.shape[1], x.shape[1]), float("nan"), device=dev)
            d.add(out, dy, x)
            res.append(out)
        for j in range(3):
            d.add(packed_out[j], packed_dy[:, 256 * j:256 * (j + 1)], packed_x[:, 256:])
        d.flush()
        outs.append([r.clone() for r in res] + [packed_out.clone()])
    for (dy, x), out in zip(probs, outs[0]):
        assert rel(out, dy.t() @ x) < 1e-5, (dy.shape, x.shape)
    for j in range(3):
        assert rel(outs[0][-1][j], packed_dy[:, 256 * j:256 * (j + 1)].t() @ packed_x[:, 256:]) < 1e-5
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


def test_grouped_column_sums(dev):
    """coda_tok_colsum_finalize_grouped_f32 through DeferredWeightGrads.add_colsum / add_split: out = sum over the partial
    rows, for the few-rows / many-columns items of the split weight gradients (four columns per thread), the many-rows
    items of the bias and LayerNorm gradients (one), outputs that are not 16-byte aligned, column counts that are not
    multiples of 4 or 256, several groups -- all in one flush, in a fixed order of additions."""
    g = torch.Generator().manual_seed(9)
    cases = [(8, 65536, 1, 0), (8, 196608, 1, 0), (16, 1000, 1, 0), (3, 260, 2, 0), (8, 1024, 1, 1), (17, 768, 1, 0),
             (256, 768, 1, 0), (40, 100, 3, 0), (1, 4, 1, 0), (12, 6, 1, 0), (8, 4096, 1, 0), (8, 4100, 1, 0)]
    outs = []
    for rep in range(2):
        d = gemm.DeferredWeightGrads(sums_only=True)
        res = []
        gen = torch.Generator().manual_seed(10)
        for blocks, n, groups, shift in cases:
            part = torch.randn(groups, blocks, n, generator=gen).to(dev)
            buf = torch.full((groups * n + 4,), float("nan"), device=dev)
            out = buf[shift:shift + groups * n].view(groups, n)     # shift 1: a 4-byte-aligned output
            d.add_colsum(part, out, blocks, n, groups)
            res.append((part, out))
        dy = torch.randn(8192, 128, generator=gen).to(dev)           # add_split: 4 chunks of 2048 rows
        x = torch.randn(8192, 256, generator=gen).to(dev)
        w = torch.full((128, 256), float("nan"), device=dev)
        d.add(w, dy, x)
        d.flush()
        for (blocks, n, groups, shift), (part, out) in zip(cases, res):
            seq = torch.zeros(groups, n, device=dev)
            for b in range(blocks):
                seq += part[:, b]
            if blocks <= 16:      # one partial row per slice: the additions are sequential in both kernels
                assert torch.equal(out, seq), (blocks, n, groups, shift)
            else:
                assert rel(out, part.double().sum(1)) < 1e-6, (blocks, n, groups)
        assert rel(w, dy.double().t() @ x.double()) < 1e-5
        outs.append([o.clone() for _, o in res] + [w.clone()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", ["f16", "f32"])
@pytest.mark.parametrize("epilogue,beta,alpha", [(0, 0.0, 1.0), (1, 0.0, 1.0), (1, 1.0, 0.5), (2, 0.0, 1.702)])
def test_gemm_ex_epilogues(dev, dtype, epilogue, beta, alpha):
    """coda_gemm_ex: C = epilogue(alpha * A B^T + beta * C + bias) in fp16 (fp32 accumulation) and fp32, with the
    first-use candidate timing switched on (the image tower's configuration) -- include/coda_gemm.h."""
    from coda_neurips2023_amd import _lib
    lib = _lib.load()
    ty = torch.float16 if dtype == "f16" else torch.float32
    g = torch.Generator().manual_seed(5)
    m, n, k = 1000, 192, 320
    a = (torch.randn(m, k, generator=g) / 4).to(dev).to(ty)
    b = (torch.randn(n, k, generator=g) / 4).to(dev).to(ty)
    c0 = torch.randn(m, n, generator=g).to(dev).to(ty)
    bias = torch.randn(n, generator=g).to(dev)
    c = c0.clone()
    for _ in range(2):     # second call: the plan from the cache (f16: the first-use-timed one)
        c.copy_(c0)
        st = lib.coda_gemm_ex(1 if dtype == "f16" else 0, epilogue, 0, 1, m, n, k, a.data_ptr(), k, b.data_ptr(), k,
                              c.data_ptr(), n, bias.data_ptr() if epilogue else None, alpha, beta,
                              _lib.current_stream_handle())
        if epilogue == 2 and st <= -3000:
            pytest.skip("the library has no swish epilogue for this problem (the tower then runs bias + its own pass)")
        _lib.check(st, "coda_gemm_ex")
    ref = alpha * (a.double() @ b.double().t()) + beta * c0.double() + (bias.double() if epilogue else 0.0)
    if epilogue == 2:
        ref = ref * torch.sigmoid(ref)
    tol = 2e-2 if dtype == "f16" else 2e-4   # half: one rounding of outputs of magnitude <= ~8 (2^-11 relative) + inputs
    assert float((c.double() - ref).abs().max()) < tol * max(1.0, float(ref.abs().max()))
    # argument checks
    assert lib.coda_gemm_ex(7, 0, 0, 1, m, n, k, a.data_ptr(), k, b.data_ptr(), k, c.data_ptr(), n, None, 1.0, 0.0, None) == -1
    assert lib.coda_gemm_ex(0, 1, 0, 1, m, n, k, a.data_ptr(), k, b.data_ptr(), k, c.data_ptr(), n, None, 1.0, 0.0, None) == -1
