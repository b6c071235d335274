"""Yardstick, not product: a plain torch.nn ResNet-50 (bottleneck v1.5, layer4 at stride 1, avg + max pooling head, BatchNorm1d neck) trained the way the
reference trains it, but with what PyTorch-ROCm gives out of the box on this GPU: eager mode, bf16 autocast, channels_last, MIOpen / hipBLASLt kernels,
torch.optim.Adam (foreach).  Batch 256 x 3 x 256 x 128, one forward + a stand-in loss + backward + Adam step per iteration.  It says what the same step
costs when nothing is hand-written; the heads / EMA of configs[1] are left out (they are < 1 ms of ours)."""
import os, sys, time
import torch, torch.nn as nn, torch.nn.functional as F

class Bottleneck(nn.Module):
    def __init__(self, cin, width, stride, down):
        super().__init__()
        self.c1 = nn.Conv2d(cin, width, 1, bias=False); self.b1 = nn.BatchNorm2d(width)
        self.c2 = nn.Conv2d(width, width, 3, stride, 1, bias=False); self.b2 = nn.BatchNorm2d(width)
        self.c3 = nn.Conv2d(width, width * 4, 1, bias=False); self.b3 = nn.BatchNorm2d(width * 4)
        self.down = nn.Sequential(nn.Conv2d(cin, width * 4, 1, stride, bias=False), nn.BatchNorm2d(width * 4)) if down else None
    def forward(self, x):
        y = F.relu(self.b1(self.c1(x))); y = F.relu(self.b2(self.c2(y))); y = self.b3(self.c3(y))
        return F.relu(y + (self.down(x) if self.down is not None else x))

class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.stem = nn.Conv2d(3, 64, 7, 2, 3, bias=False); self.bn = nn.BatchNorm2d(64)
        layers, cin = [], 64
        for width, n, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 1)):
            for i in range(n):
                layers.append(Bottleneck(cin, width, stride if i == 0 else 1, i == 0)); cin = width * 4
        self.trunk = nn.Sequential(*layers)
        self.neck = nn.BatchNorm1d(2048)
    def forward(self, x):
        x = F.max_pool2d(self.bn(self.stem(x)), 3, 2, 1)
        x = self.trunk(x)
        return self.neck(F.adaptive_avg_pool2d(x, 1).flatten(1) + F.adaptive_max_pool2d(x, 1).flatten(1))

dev = torch.device("cuda")
net = Net().to(dev).to(memory_format=torch.channels_last).train()
opt = torch.optim.Adam(net.parameters(), lr=3.5e-4, weight_decay=5e-4)
x = torch.randn(256, 3, 256, 128, device=dev).contiguous(memory_format=torch.channels_last)
tgt = F.normalize(torch.randn(256, 2048, device=dev), dim=1)
AUTOCAST = os.environ.get("AUTOCAST", "1") != "0"      # 0: fp32, what the reference's own script runs (mainKIT.py has no autocast)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=AUTOCAST):
        emb = net(x)
    loss = (1 - (F.normalize(emb.float(), dim=1) * tgt).sum(1)).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
warm, steps = int(os.environ.get("WARM", "6")), int(os.environ.get("STEPS", "10"))
for i in range(warm):
    t0 = time.perf_counter(); step(); torch.cuda.synchronize()
    print("warm-up step %d: %.1f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps): step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
print("torch eager %s channels_last ResNet-50 (batch 256, 256x128): %.2f ms/step = %.0f images/s  [MIOPEN_FIND_MODE=%s]" % ("bf16 autocast" if AUTOCAST else "fp32", ms, 256 / ms * 1e3, os.environ.get("MIOPEN_FIND_MODE", "default")))
