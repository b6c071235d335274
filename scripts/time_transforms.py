"""Throughput of the GPU image transform (resize + train augmentation) vs the PIL oracle on the host (one core)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daliid_amd import transforms as T
from oracle import augment as OA
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, size=(128, 64, 3), dtype=np.uint8) for _ in range(256)]          # Market-1501 crops are 128 x 64
torch.manual_seed(0)
params = T.sample_train_params(256, 256, 128)
# device-resident inputs: time the two kernels alone
u8 = T.resize_bicubic_u8(imgs, 256, 128)
prm = torch.from_numpy(params).cuda()
for _ in range(3): T.augment(u8, prm)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): out = T.augment(u8, prm)
e1.record(); torch.cuda.synchronize()
aug_ms = e0.elapsed_time(e1) / 20
t0 = time.perf_counter()
for _ in range(5): u8 = T.resize_bicubic_u8(imgs, 256, 128)
torch.cuda.synchronize()
rs_ms = (time.perf_counter() - t0) / 5 * 1e3
t0 = time.perf_counter()
for i in range(64): OA.train_transform(OA.resize(imgs[i], 256, 128), params[i])
cpu_ms = (time.perf_counter() - t0) / 64 * 1e3
print("batch 256 -> 256x128: augment kernel %.3f ms (%.0f img/s, %.2f TB/s of 15 B/px); resize incl. host packing + H2D %.2f ms; "
      "PIL on one host core %.2f ms per image (%.0f img/s)" % (aug_ms, 256 / aug_ms * 1e3, 256 * 256 * 128 * 15 / aug_ms / 1e9, rs_ms, cpu_ms, 1e3 / cpu_ms))
