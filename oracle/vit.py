"""Oracle: TransReID ViT forward as a pure function of a reference-keyed state_dict (fp32 CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against the reference's own
``vit_pytorch.TransReID`` / ``make_models.build_transformer`` by tests/golden/vit_*.npz.

Restates vit_pytorch.py:375-408 (forward_features, camera=view=0, local_feature=False),
Block (:167-184: pre-LN residual), Attention (:139-164: qkv Linear, softmax(q k^T * hd^-0.5) v,
proj), Mlp (:120-136: fc1, exact-erf GELU, fc2), PatchEmbed_overlap (:251-288: conv
patch x patch / stride), final LayerNorm and cls token; and the BN1d neck of
make_models.build_transformer.forward (make_models.py:184-205).  DropPath/Dropout are
identity here (parity configs run with rate 0).
"""
import torch
import torch.nn.functional as F


def transreid_forward(sd, x, num_heads, patch=16, stride=16, ln_eps=1e-6, prefix=""):
    """sd: dict with reference key names (optionally under ``prefix`` e.g. 'base.').
    x [B,3,H,W] -> cls feature [B,dim]."""
    g = lambda k: sd[prefix + k]
    B = x.shape[0]
    t = F.conv2d(x, g("patch_embed.proj.weight"), g("patch_embed.proj.bias"), stride=stride)
    t = t.flatten(2).transpose(1, 2)                                   # [B, N, dim]
    t = torch.cat((g("cls_token").expand(B, -1, -1), t), dim=1) + g("pos_embed")
    dim = t.shape[-1]
    hd = dim // num_heads
    depth = 0
    while (prefix + "blocks.%d.norm1.weight" % depth) in sd:
        depth += 1
    for i in range(depth):
        p = "blocks.%d." % i
        h = F.layer_norm(t, (dim,), g(p + "norm1.weight"), g(p + "norm1.bias"), ln_eps)
        qkv = F.linear(h, g(p + "attn.qkv.weight"), sd.get(prefix + p + "attn.qkv.bias"))
        N = t.shape[1]
        qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * (hd ** -0.5)).softmax(dim=-1)
        h = (attn @ v).transpose(1, 2).reshape(B, N, dim)
        t = t + F.linear(h, g(p + "attn.proj.weight"), g(p + "attn.proj.bias"))
        h = F.layer_norm(t, (dim,), g(p + "norm2.weight"), g(p + "norm2.bias"), ln_eps)
        h = F.gelu(F.linear(h, g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias")))
        t = t + F.linear(h, g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias"))
    t = F.layer_norm(t, (dim,), g("norm.weight"), g("norm.bias"), ln_eps)
    return t[:, 0]


def build_transformer_forward(sd, x, num_heads, patch=16, stride=16, training=False, bn_eps=1e-5):
    """make_models.py:184-205: feat = BatchNorm1d(base(x)).  In training mode uses batch
    statistics (running-stat update not modelled here: the caller compares ``feat`` only)."""
    gf = transreid_forward(sd, x, num_heads, patch, stride, prefix="base.")
    return F.batch_norm(gf, sd["bottleneck.running_mean"].clone(), sd["bottleneck.running_var"].clone(),
                        sd["bottleneck.weight"], sd["bottleneck.bias"], training, 0.1, bn_eps)
