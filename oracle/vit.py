"""Oracle: TransReID ViT forward as a pure function of a reference-keyed state_dict (fp32 CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against the reference's own
``vit_pytorch.TransReID`` / ``make_models.build_transformer`` by tests/golden/vit_*.npz.

Restates vit_pytorch.py:375-408 (forward_features, camera=view=0, local_feature=False),
Block (:167-184: pre-LN residual), Attention (:139-164: qkv Linear, softmax(q k^T * hd^-0.5) v,
proj), Mlp (:120-136: fc1, exact-erf GELU, fc2), PatchEmbed_overlap (:251-288: conv
patch x patch / stride), final LayerNorm and cls token; and the BN1d neck of
make_models.build_transformer.forward (make_models.py:184-205).  Dropout is identity (rate 0
in every reference caller).  DropPath (vit_pytorch.py:45-62, per-block rate
``linspace(0, rate, depth)`` :338) is restated with the uniform draws passed IN
(``drop_path=(rate, u[2*depth, B])``), so a test can hand the same draws to the HIP plan;
pinned by tests/golden/vit_droppath.npz (the reference's own train-mode forward/backward
with the draws it consumed).
"""
import torch
import torch.nn.functional as F


def _drop_path(branch, rate_i, u):
    """vit_pytorch.py:54-61 with the uniform draw u [B] given: x.div(keep) * floor(keep + u); rate 0 = nn.Identity (:171)."""
    if rate_i == 0.0:
        return branch
    keep = 1.0 - rate_i
    random_tensor = (keep + u.reshape(-1, 1, 1).to(branch.dtype)).floor()
    return branch.div(keep) * random_tensor


def transreid_forward(sd, x, num_heads, patch=16, stride=16, ln_eps=1e-6, prefix="", drop_path=None):
    """sd: dict with reference key names (optionally under ``prefix`` e.g. 'base.').
    x [B,3,H,W] -> cls feature [B,dim].  drop_path = (rate, u [2*depth, B]) applies training-mode DropPath."""
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
    dpr = [v.item() for v in torch.linspace(0, drop_path[0], depth)] if drop_path is not None else [0.0] * depth      # :338
    for i in range(depth):
        p = "blocks.%d." % i
        u_att = drop_path[1][2 * i] if drop_path is not None else None
        u_mlp = drop_path[1][2 * i + 1] if drop_path is not None else None
        h = F.layer_norm(t, (dim,), g(p + "norm1.weight"), g(p + "norm1.bias"), ln_eps)
        qkv = F.linear(h, g(p + "attn.qkv.weight"), sd.get(prefix + p + "attn.qkv.bias"))
        N = t.shape[1]
        qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * (hd ** -0.5)).softmax(dim=-1)
        h = (attn @ v).transpose(1, 2).reshape(B, N, dim)
        t = t + _drop_path(F.linear(h, g(p + "attn.proj.weight"), g(p + "attn.proj.bias")), dpr[i], u_att)
        h = F.layer_norm(t, (dim,), g(p + "norm2.weight"), g(p + "norm2.bias"), ln_eps)
        h = F.gelu(F.linear(h, g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias")))
        t = t + _drop_path(F.linear(h, g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias")), dpr[i], u_mlp)
    t = F.layer_norm(t, (dim,), g("norm.weight"), g("norm.bias"), ln_eps)
    return t[:, 0]


def build_transformer_forward(sd, x, num_heads, patch=16, stride=16, training=False, bn_eps=1e-5, drop_path=None):
    """make_models.py:184-205: feat = BatchNorm1d(base(x)).  In training mode uses batch
    statistics (running-stat update not modelled here: the caller compares ``feat`` only)."""
    gf = transreid_forward(sd, x, num_heads, patch, stride, prefix="base.", drop_path=drop_path if training else None)
    return F.batch_norm(gf, sd["bottleneck.running_mean"].clone(), sd["bottleneck.running_var"].clone(),
                        sd["bottleneck.weight"], sd["bottleneck.bias"], training, 0.1, bn_eps)
