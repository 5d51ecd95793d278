"""The frozen CLIP image tower of the distillation branch.

Mirror of the reference's ``VisionTransformer`` (CLIP/clip/model.py:595-659) and of the part of its ``CLIP`` class
the detector touches (``.visual``, ``.dtype``, ``.encode_image``, :1056-1067): same constructor arguments, same
parameter names -- an OpenAI CLIP ViT checkpoint's ``visual.*`` entries load with ``load_state_dict`` -- and the
same return value, ``(class-token embedding (n, out), all-token embeddings (n, L, out))``.  The body is one call
into ``libcoda_hip.so`` (``coda_vit_fwd``, include/coda_clip_tower.h): hipBLASLt GEMMs with the bias / residual /
QuickGELU work in their epilogues and this library's kernels for LayerNorm, token assembly and attention.
Inference only -- the tower is frozen in the reference (models/model_3detr.py:331-333) -- and GPU only.

The compute type follows the weights, as in the reference (``CLIP.dtype`` = ``visual.conv1.weight.dtype``):
``convert_weights`` / ``.half()`` gives the fp16 tower ``clip.load`` builds on a GPU, float32 weights give the
float32 tower (the parity mode of the tests).

The text tower is not part of the hot path (its 512-d class embeddings are computed once at start-up,
models/model_3detr.py:339-342) and stays with the deployment's CLIP module.
"""
import ctypes
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib


class LayerNorm(nn.LayerNorm):
    """Parameter holder; the fused tower normalises in float32 whatever the activation type (:254-260)."""


class QuickGELU(nn.Module):
    """Parameter-free marker of the ``mlp.gelu`` slot (the tower applies it as the c_fc GEMM's epilogue)."""

    def forward(self, u):
        return u / (1.0 + torch.exp(-1.702 * u))


class ResidualAttentionBlock(nn.Module):
    """Parameter layout of CLIP/clip/model.py:295-311 (``attn.in_proj_weight`` ... ``mlp.c_proj.bias``)."""

    def __init__(self, d_model, n_head, attn_mask=None):
        super().__init__()
        if attn_mask is not None:
            raise NotImplementedError("the image tower has no attention mask (CLIP/clip/model.py:607)")
        hidden = 4 * d_model
        self.attn = nn.MultiheadAttention(embed_dim=d_model, num_heads=n_head)
        self.ln_1, self.ln_2 = LayerNorm(d_model), LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, hidden)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(hidden, d_model))]))


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, attn_mask=None):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        blocks = [ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)]
        self.resblocks = nn.Sequential(*blocks)


class _Layer(ctypes.Structure):
    _fields_ = [(name, ctypes.c_void_p) for name in
                ("ln1_g", "ln1_b", "ln2_g", "ln2_b", "in_w", "in_b", "out_w", "out_b", "fc_w", "fc_b", "proj_w", "proj_b")]


class _Desc(ctypes.Structure):
    _fields_ = ([(name, ctypes.c_int32) for name in
                 ("dtype", "resolution", "patch", "width", "nlayers", "heads", "mlp", "out_dim")]
                + [("eps", ctypes.c_float), ("pad_", ctypes.c_int32)]
                + [(name, ctypes.c_void_p) for name in
                   ("conv_w", "cls", "pos", "ln_pre_g", "ln_pre_b", "ln_post_g", "ln_post_b", "proj", "layers")])


_DTYPE_CODE = {torch.float32: 0, torch.float16: 1}


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution, self.patch_size, self.output_dim = input_resolution, patch_size, output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        tokens = (input_resolution // patch_size) ** 2 + 1

        def init(*shape):  # placeholder values: the deployment loads the CLIP checkpoint over them
            return nn.Parameter(torch.randn(*shape) / width ** 0.5)

        self.class_embedding, self.positional_embedding = init(width), init(tokens, width)
        self.ln_pre, self.ln_post = LayerNorm(width), LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.proj = init(width, output_dim)
        self._packed = None
        for p in self.parameters():  # frozen: inference only
            p.requires_grad_(False)

    # ---- weights as the C driver wants them: matrices in the compute type, vectors in float32 -------------------
    def refresh(self):
        """Drop the converted weight copies: the next call converts again.  The cache below is keyed on (pointer,
        version counter, dtype) of every parameter, which ``load_state_dict`` / ``.to()`` / in-place tensor ops
        change; a write through ``p.data`` (``p.data.copy_(...)``, as some checkpoint loaders do) changes neither, so
        such loaders call this afterwards.  ``load_state_dict`` and ``_apply`` (``.to()`` / ``.half()`` / ``.cuda()``) do."""
        self._packed = None

    def _load_from_state_dict(self, *args, **kwargs):
        self._packed = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._packed = None
        return super()._apply(fn, *args, **kwargs)

    def _pack(self):
        params = list(self.parameters())
        stamp = tuple((p.data_ptr(), p._version, p.dtype) for p in params)
        if self._packed is not None and self._packed[0] == stamp:
            return self._packed[1]
        ty = self.conv1.weight.dtype
        if ty not in _DTYPE_CODE:
            raise RuntimeError(f"image tower: weights must be float16 or float32, got {ty}")
        keep = []

        def mat(t):
            keep.append(t.detach().to(ty).contiguous())
            return keep[-1].data_ptr()

        def vec(t):
            keep.append(t.detach().to(torch.float32).contiguous())
            return keep[-1].data_ptr()

        blocks = list(self.transformer.resblocks)
        layers = (_Layer * max(len(blocks), 1))()
        for i, blk in enumerate(blocks):
            if blk.ln_1.eps != self.ln_pre.eps or blk.ln_2.eps != self.ln_pre.eps:
                raise RuntimeError("image tower: one LayerNorm epsilon for all layers")
            layers[i] = _Layer(vec(blk.ln_1.weight), vec(blk.ln_1.bias), vec(blk.ln_2.weight), vec(blk.ln_2.bias),
                               mat(blk.attn.in_proj_weight), vec(blk.attn.in_proj_bias),
                               mat(blk.attn.out_proj.weight), vec(blk.attn.out_proj.bias),
                               mat(blk.mlp.c_fc.weight), vec(blk.mlp.c_fc.bias),
                               mat(blk.mlp.c_proj.weight), vec(blk.mlp.c_proj.bias))
        width = self.conv1.weight.shape[0]
        desc = _Desc(_DTYPE_CODE[ty], self.input_resolution, self.patch_size, width, len(blocks),
                     self.transformer.heads, blocks[0].mlp.c_fc.weight.shape[0] if blocks else 4 * width,
                     self.proj.shape[1], float(self.ln_pre.eps), 0,
                     mat(self.conv1.weight.reshape(width, -1)), vec(self.class_embedding),
                     vec(self.positional_embedding), vec(self.ln_pre.weight), vec(self.ln_pre.bias),
                     vec(self.ln_post.weight), vec(self.ln_post.bias), mat(self.proj),
                     ctypes.cast(layers, ctypes.c_void_p).value)
        self._packed = (stamp, (desc, layers, keep, ty))
        return self._packed[1]

    @torch.no_grad()
    def embed(self, x, tokens=True):
        """x (n,3,res,res) -> class-token embeddings (n, out) [, all tokens (n, L, out)] in the weights' type."""
        if not x.is_cuda:
            raise RuntimeError("CPU not supported")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.input_resolution or x.shape[3] != self.input_resolution:
            raise RuntimeError(f"image tower expects (n,3,{self.input_resolution},{self.input_resolution}), got {tuple(x.shape)}")
        if self.conv1.weight.device != x.device:
            raise RuntimeError("image tower: weights and images are on different devices")
        desc, _layers, _keep, ty = self._pack()
        lib = _lib.load()
        n = x.shape[0]
        img = x.detach().to(torch.float32).contiguous()
        ntok = (self.input_resolution // self.patch_size) ** 2 + 1
        cls = torch.empty((n, self.output_dim), dtype=ty, device=x.device)
        tok = torch.empty((ntok, n, self.output_dim), dtype=ty, device=x.device) if tokens else None
        with torch.cuda.device(x.device):
            nbytes = lib.coda_vit_workspace_bytes(ctypes.byref(desc), n, int(tokens))
            if nbytes == 0 and n > 0:
                raise _lib.CodaLibraryError("coda_vit_workspace_bytes: unsupported tower configuration")
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=x.device)
            st = lib.coda_vit_fwd(ctypes.byref(desc), img.data_ptr(), n, cls.data_ptr(),
                                  tok.data_ptr() if tokens else None, ws.data_ptr(), nbytes,
                                  _lib.current_stream_handle())
        _lib.check(st, "coda_vit_fwd")
        return (cls, tok.permute(1, 0, 2)) if tokens else cls

    def forward(self, x, im_name=None, max_w=None, if_pool=True, if_early_feat=False):
        """The reference's call (CLIP/clip/model.py:612); the debugging arguments are accepted and unused, as there."""
        return self.embed(x, tokens=True)


class ImageTower(nn.Module):
    """What the detector uses of the reference's ``CLIP`` object on the image side: ``.visual``, ``.dtype``,
    ``.encode_image`` (CLIP/clip/model.py:1056-1067).  ``tokens=False`` (default) skips the all-token projection
    the distillation branch throws away (models/model_3detr.py:1097-1098 keeps element 0 of the tuple)."""

    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size):
        super().__init__()
        self.visual = VisionTransformer(image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_width // 64, embed_dim)

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image, im_name=None, max_w=None, if_pool=True, if_early_feat=False, tokens=False):
        return self.visual.embed(image, tokens=tokens)


def convert_weights(model):
    """fp16 matrices, float32 LayerNorm parameters / embeddings: what CLIP/clip/model.py:1146-1167 does to the tower."""
    for m in model.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            m.weight.data = m.weight.data.half()
            if m.bias is not None:
                m.bias.data = m.bias.data.half()
        if isinstance(m, nn.MultiheadAttention):
            m.in_proj_weight.data = m.in_proj_weight.data.half()
            m.in_proj_bias.data = m.in_proj_bias.data.half()
        if isinstance(m, VisionTransformer):
            m.proj.data = m.proj.data.half()
    return model


def build_image_tower(state_dict, half=True):
    """An ``ImageTower`` from a CLIP checkpoint's state dict (the ``visual.*`` entries; sizes are read off the
    tensors as CLIP/clip/model.py:1267-1274 does).  ResNet towers (``visual.layer1...``) are not supported: the
    reference's recipes load ViT-B/16 (models/model_3detr.py:325)."""
    if "visual.proj" not in state_dict:
        raise RuntimeError("build_image_tower: not a ViT checkpoint (no 'visual.proj')")
    width = state_dict["visual.conv1.weight"].shape[0]
    layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    patch = state_dict["visual.conv1.weight"].shape[-1]
    grid = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    model = ImageTower(state_dict["visual.proj"].shape[1], patch * grid, layers, width, patch)
    visual = {k[len("visual."):]: v for k, v in state_dict.items() if k.startswith("visual.")}
    model.visual.load_state_dict(visual, strict=True)
    return convert_weights(model) if half else model.float()
