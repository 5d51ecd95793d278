"""The module-by-module transformer path (what runs for every configuration the fused GPU path does not
cover) against the reference fixture, on the CPU: oracle/cpu_port.py supplies a plain torch attention core and
the C oracle ops for the interim set abstraction; the layer / stack code under test is the product's."""
import pytest
import torch

from oracle import cpu_port
from tests import test_transformer_gpu as G


@pytest.fixture(autouse=True)
def _fixture_mode(_distance_mode_default):
    from tests._modes import fixture_mode, set_distance_mode
    set_distance_mode(fixture_mode(G.G))


@pytest.mark.parametrize("check", [G.test_encoder_stack, G.test_masked_encoder_with_interim_downsampling,
                                   G.test_decoder_stack_with_attention_weights, G.test_single_layers_real_width],
                         ids=lambda f: f.__name__)
def test_module_path_matches_reference_fixture_on_cpu(check):
    with cpu_port.patched():
        check(torch.device("cpu"))


def test_post_norm_layers_follow_the_reference_order():
    """normalize_before=False: x = norm(x + block(x)) per sub-layer in the decoder; in the encoder layer the
    attention block stays un-normed (the reference only norms it under a flag it never sets)."""
    from coda_neurips2023_amd import transformer as T
    torch.manual_seed(0)
    with cpu_port.patched():
        el = T.TransformerEncoderLayer(64, nhead=4, dim_feedforward=32, dropout=0.0, normalize_before=False).eval()
        x = torch.randn(10, 2, 64)
        a = x + el.self_attn(x, x, value=x, need_weights=False)[0]
        f = el.linear2(torch.relu(el.linear1(a)))
        assert torch.allclose(el(x), el.norm2(a + f), atol=1e-6)
        dl = T.TransformerDecoderLayer(64, nhead=4, dim_feedforward=32, dropout=0.0, normalize_before=False).eval()
        t, m = torch.randn(6, 2, 64), torch.randn(10, 2, 64)
        s1 = dl.norm1(t + dl.self_attn(t, t, value=t, need_weights=False)[0])
        s2 = dl.norm2(s1 + dl.multihead_attn(s1, m, value=m, need_weights=False)[0])
        s3 = dl.norm3(s2 + dl.linear2(torch.relu(dl.linear1(s2))))
        out, attn = dl(t, m)
        assert attn is None and torch.allclose(out, s3, atol=1e-6)
