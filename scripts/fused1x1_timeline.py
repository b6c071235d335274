"""Where a workgroup of the persistent streaming kernel (csrc/fused1x1.h) spends its time: in-kernel s_memrealtime sums per phase (diagnostic).
   python scripts/fused1x1_timeline.py [batch=256]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daliid_amd import _lib, ops_nn as nn
bf16 = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
os.environ.setdefault("DALI_CONV_PERSIST_KMAX", "512")            # the diagnostic covers layer4 too (the plan keeps it on the tile-per-workgroup kernel)
L = _lib.lib(); L.dali_debug_set_conv_stamps.argtypes = [ctypes.c_void_p]
for name, P, K, Cm in [("layer1", B * 2048, 64, 256), ("layer2", B * 512, 128, 512), ("layer3", B * 128, 256, 1024), ("layer4", B * 128, 512, 2048)]:
    x = torch.randn(P, K, device="cuda").to(bf16); w = (torch.randn(Cm, K, device="cuda") / K ** 0.5).to(bf16)
    res = torch.randn(P, Cm, device="cuda").to(bf16)
    sc, sh = torch.rand(Cm, device="cuda") + 0.5, torch.randn(Cm, device="cuda")
    run = lambda: nn.conv1x1_fused(x, w, sc, sh, None, res, relu=True, want_bits=True)
    for _ in range(3): run()
    stamps = torch.zeros(1 << 12, 12, device="cuda", dtype=torch.int64)
    L.dali_debug_set_conv_stamps(ctypes.c_void_p(stamps.data_ptr()))
    run(); torch.cuda.synchronize()
    L.dali_debug_set_conv_stamps(None)
    s = stamps.cpu().numpy().astype(np.float64)
    s = s[s[:, 3] > 0]
    T = s[:, 8]
    us = lambda col: (s[:, col] * 0.01 / T).mean()              # 100 MHz ticks -> us per tile
    life = ((s[:, 3] - s[:, 0]) * 0.01)
    print("%s K=%d Cm=%d: %d workgroups, %.1f tiles each, lifetime %.1f us (max %.1f) = %.2f us per tile" % (name, K, Cm, len(s), T.mean(), life.mean(), life.max(), (life / T).mean()))
    print("   consumer per tile: main loop work %.2f + barrier wait %.2f | stage 1 work %.2f + wait %.2f | stage 2 (stores) work %.2f + wait %.2f"
          % (us(1), us(2), us(4), us(5), us(6), us(7)))
    print("   ring producer 0 per tile: issue %.2f, vmcnt wait %.2f, barrier wait %.2f (main beats only)" % (us(9), us(10), us(11)))
