"""tuning.py on the GPU box: the shipped table is accepted (same library versions as it was measured with), TunableOp
runs in look-up mode, and products of listed shapes are still the products."""
import pytest
import torch

from coda_neurips2023_amd import tuning

pytestmark = pytest.mark.gpu


def test_table_is_in_use_and_products_are_products(dev):
    import torch.cuda.tunable as tunable
    if "gfx950" not in getattr(torch.cuda.get_device_properties(dev), "gcnArchName", ""):
        pytest.skip("table is for gfx950")
    on = tuning.enable_tuned_gemms()
    val = dict((v[0], v[1]) for v in tunable.get_validators())
    table = tuning.table_validators()
    if any(table.get(k) != val.get(k) for k in ("ROCBLAS_VERSION", "HIPBLASLT_VERSION", "PT_VERSION")):
        assert not on          # other library versions: the table must have been refused
        pytest.skip("table measured with other library versions")
    assert on and tuning.is_on() and tunable.is_enabled() and not tunable.tuning_is_enabled()
    g = torch.Generator(device="cpu").manual_seed(5)
    for (m, k, n) in [(16384, 256, 256), (327680, 64, 128), (2048, 256, 256)]:      # nn_256_16384_256, nn_128_P_64-like ...
        a = torch.randn(m, k, generator=g).to(dev)
        b = torch.randn(k, n, generator=g).to(dev)
        got = a @ b
        ref = (a.double() @ b.double())
        err = float((got.double() - ref).abs().max() / ref.abs().max())
        assert err < 1e-5, (m, k, n, err)
    x = torch.randn(327680, 64, generator=g).to(dev)
    w = torch.randn(128, 64, generator=g).to(dev)
    got = torch.mm(x, w.t())                                                        # tn_128_327680_64
    assert float((got.double() - x.double() @ w.double().t()).abs().max()) < 1e-3
