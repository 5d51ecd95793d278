"""The one-launch Fourier coordinate embedding (csrc/pos_embed.hip) against the module's own torch expression -- the
reference's operation sequence (models/position_embedding.py:97-130) -- and against float64."""
import math

import pytest
import torch

from coda_neurips2023_amd.position_embedding import PositionEmbeddingCoordsSine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("b,n,d_pos,channels,normalize", [(8, 2048, 256, None, True), (8, 256, 256, None, True),
                                                          (3, 77, 128, 64, True), (2, 50, 256, None, False),
                                                          (1, 1, 32, None, True)])
def test_fourier_embedding_kernel_matches_the_torch_expression(dev, b, n, d_pos, channels, normalize):
    torch.manual_seed(b * 1000 + n)
    emb = PositionEmbeddingCoordsSine(d_pos=d_pos, pos_type="fourier", normalize=normalize).to(dev)
    xyz = (torch.rand(b, n, 3, device=dev) * 6 - 3)
    lo = xyz.amin(1) - torch.rand(b, 3, device=dev)
    hi = xyz.amax(1) + torch.rand(b, 3, device=dev)
    rng = [lo, hi]
    got = emb(xyz, num_channels=channels, input_range=rng)                       # no_grad inside: the kernel
    with torch.enable_grad():
        ref = emb.get_fourier_embeddings(xyz, channels, rng)                     # grad mode: the torch expression
    assert got.shape == ref.shape and got.stride() == ref.stride()               # (B, C, N) view of a (B, N, C) buffer
    unit = (xyz.double() - lo.double()[:, None]) / (hi.double() - lo.double())[:, None] if normalize else xyz.double()
    half = got.shape[1] // 2
    phase = (unit * (2 * math.pi)) @ emb.gauss_B[:, :half].double()
    exact = torch.cat((phase.sin(), phase.cos()), 2).transpose(1, 2)
    err_kernel = float((got.double() - exact).abs().max())
    err_torch = float((ref.double() - exact).abs().max())
    assert err_kernel <= max(2 * err_torch, 2e-6), (err_kernel, err_torch)
    assert float((got - ref).abs().max()) < 1e-5


def test_fourier_embedding_kernel_rejects_nothing_it_should_take(dev):
    emb = PositionEmbeddingCoordsSine(d_pos=64, pos_type="fourier", normalize=True).to(dev)
    xyz = torch.zeros(2, 0, 3, device=dev)
    out = emb(xyz, input_range=[torch.zeros(2, 3, device=dev), torch.ones(2, 3, device=dev)])
    assert out.shape == (2, 64, 0)
