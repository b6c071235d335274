"""Yardstick, not product: a plain torch.nn ViT-B/16 (12 blocks, 768 wide, 12 heads, 197 tokens at 224 x 224, DropPath 0.1, BatchNorm1d neck) under
PyTorch-ROCm out of the box: eager, bf16 autocast, F.scaled_dot_product_attention, torch.optim.Adam.  Batch 128, forward + stand-in loss + backward + Adam."""
import os, time
import torch, torch.nn as nn, torch.nn.functional as F

class Block(nn.Module):
    def __init__(self, d=768, h=12, dp=0.0):
        super().__init__()
        self.n1, self.n2 = nn.LayerNorm(d, eps=1e-6), nn.LayerNorm(d, eps=1e-6)
        self.qkv, self.proj = nn.Linear(d, 3 * d), nn.Linear(d, d)
        self.fc1, self.fc2 = nn.Linear(d, 4 * d), nn.Linear(4 * d, d)
        self.h, self.dp = h, dp
    def drop_path(self, x):
        if not self.training or self.dp == 0.0: return x
        keep = 1 - self.dp
        m = x.new_empty(x.shape[0], 1, 1).bernoulli_(keep)
        return x * m / keep
    def forward(self, x):
        B, T, D = x.shape
        q, k, v = self.qkv(self.n1(x)).reshape(B, T, 3, self.h, D // self.h).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, T, D)
        x = x + self.drop_path(self.proj(a))
        return x + self.drop_path(self.fc2(F.gelu(self.fc1(self.n2(x)))))

class ViT(nn.Module):
    def __init__(self):
        super().__init__()
        self.pe = nn.Conv2d(3, 768, 16, 16)
        self.cls = nn.Parameter(torch.zeros(1, 1, 768)); self.pos = nn.Parameter(torch.zeros(1, 197, 768))
        self.blocks = nn.Sequential(*[Block(dp=0.1 * i / 11) for i in range(12)])
        self.norm = nn.LayerNorm(768, eps=1e-6); self.neck = nn.BatchNorm1d(768)
    def forward(self, x):
        x = self.pe(x).flatten(2).transpose(1, 2)
        x = torch.cat((self.cls.expand(x.shape[0], -1, -1), x), 1) + self.pos
        return self.neck(self.norm(self.blocks(x))[:, 0])

dev = torch.device("cuda")
net = ViT().to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=3.5e-4, weight_decay=5e-4)
x = torch.randn(128, 3, 224, 224, device=dev)
tgt = F.normalize(torch.randn(128, 768, device=dev), dim=1)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        emb = net(x)
    loss = (1 - (F.normalize(emb.float(), dim=1) * tgt).sum(1)).mean()
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for i in range(5):
    t0 = time.perf_counter(); step(); torch.cuda.synchronize()
    print("warm-up step %d: %.1f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("torch eager bf16 ViT-B/16 (batch 128, 224x224, SDPA): %.2f ms/step = %.0f images/s" % (ms, 128 / ms * 1e3))
