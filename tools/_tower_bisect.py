import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coda_neurips2023_amd import _lib, clip_tower
lib = _lib.load()
which = sys.argv[1]
if which == "attn":
    n = int(sys.argv[2])
    qkv = torch.randn(197, n, 2304, device="cuda").half()
    out = torch.empty(197, n, 768, dtype=torch.float16, device="cuda")
    print("attn", n, lib.coda_vit_attention_f16(qkv.data_ptr(), out.data_ptr(), n, 197, 12, None)); torch.cuda.synchronize(); print("ok", float(out.float().abs().max()))
elif which == "gemm":
    m, n, k, epi, beta = (int(v) for v in sys.argv[2:7])
    a = torch.randn(m, k, device="cuda").half(); b = torch.randn(n, k, device="cuda").half()
    c = torch.zeros(m, n, device="cuda").half(); bias = torch.randn(n, device="cuda")
    st = lib.coda_gemm_ex(1, epi, 0, 1, m, n, k, a.data_ptr(), k, b.data_ptr(), k, c.data_ptr(), n, bias.data_ptr() if epi else None, 1.0, float(beta), None)
    torch.cuda.synchronize(); print("gemm", m, n, k, epi, beta, st)
    if os.environ.get("TORCH_TOO"):
        y = torch.nn.functional.linear(a, b, bias.half()); torch.cuda.synchronize(); print("torch ok", float((y.float()-c.float()).abs().max()))
    ref = a.float() @ b.float().t() + (bias if epi else 0)
    if epi == 2: ref = ref * torch.sigmoid(ref)
    print("err", float((c.float() - ref).abs().max()), float(ref.abs().max()))
else:
    n = int(sys.argv[2])
    tower = clip_tower.convert_weights(clip_tower.ImageTower(512, 224, 12, 768, 16)).cuda()
    x = torch.randn(n, 3, 224, 224, device="cuda")
    y = tower.encode_image(x); torch.cuda.synchronize(); print("tower", n, float(y.float().abs().max()))
