"""Inference forward of the ResNet-50-ReID plan at batch B (extractFeatures forwards the train set / gallery at 500): ms per batch, images / s.
   python scripts/time_eval_forward.py [B=500] [reps=20]      (DALI_EVAL_FUSED=0: the training dataflow, for A/B)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliid_amd import Encoders
B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
net = Encoders.ResNet50ReID(seed=12).eval()
x = torch.randn(B, 3, 256, 128, device="cuda")
with torch.no_grad():
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        net(x)
    ev[1].record()
    torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / n
print("eval forward B=%d fused=%s: %.3f ms per batch -> %.0f img/s, %.1f TFLOP/s (8.107 GF/img)"
      % (B, os.environ.get("DALI_EVAL_FUSED", "1"), ms, B / ms * 1e3, B * 8.107e9 / (ms * 1e-3) / 1e12))
