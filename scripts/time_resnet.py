"""Quick timing of the ResNet-50-ReID plan forward+backward at batch B (debug aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliid_amd import Encoders
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
net = Encoders.ResNet50ReID().train()
x = torch.randn(B, 3, 256, 128, device="cuda")
d = torch.randn(B, 2048, device="cuda")
def step():
    emb = net._run_forward(x, True)
    for s in range(4):
        net._backward_stage(d, s)
for _ in range(3): step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
n = 10
ev[0].record()
for _ in range(n): net._run_forward(x, True)
ev[1].record()
for _ in range(n): step()
ev[2].record()
torch.cuda.synchronize()
f = ev[0].elapsed_time(ev[1]) / n; fb = ev[1].elapsed_time(ev[2]) / n
print("B=%d fwd %.2f ms, fwd+bwd %.2f ms -> %.0f img/s, %.1f TFLOP/s (24.32 GF/img)" % (B, f, fb, B / fb * 1e3, B * 24.32e9 / (fb * 1e-3) / 1e12))
print("arena GB", net._arena.numel() / 1e9)
