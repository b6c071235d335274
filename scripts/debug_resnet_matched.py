"""HIP plan vs rounding-matched oracle vs fp32 oracle: embeddings and parameter gradients (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy
import numpy as np
import torch
from oracle.resnet50_reid import ResNet50ReID as OracleNet
from oracle.resnet50_bf16 import forward_matched
from daliid_amd import Encoders

def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))

if len(sys.argv) > 1 and sys.argv[1] == "full":
    layers, width, shape = (3, 4, 6, 3), 64, (32, 3, 256, 128)
elif len(sys.argv) > 1 and sys.argv[1] == "tiny":
    layers, width, shape = (1, 1, 1, 1), 32, (int(os.environ.get("B", "32")), 3, 128, 64)
else:
    layers, width, shape = (2, 1, 2, 1), 64, (5, 3, 96, 48)
gam = float(os.environ.get("BN3_GAMMA", "1.0"))
torch.manual_seed(2)
ref = OracleNet(layers=layers, width=width)
g = torch.Generator().manual_seed(3)
with torch.no_grad():
    for m in ref.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
            m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
    for m in ref.modules():
        if hasattr(m, "bn3"):
            m.bn3.weight.mul_(gam)
ref2 = copy.deepcopy(ref)
net = Encoders.ResNet50ReID(layers=layers, width=width)
net.load_state_dict(ref.state_dict())
x = torch.randn(*shape, generator=g)
if os.environ.get("STRUCT", "0") == "1":
    low = torch.randn(shape[0], 3, shape[2] // 16, shape[3] // 16, generator=g)
    x = 2.0 * torch.nn.functional.interpolate(low, size=shape[2:], mode="bilinear", align_corners=False) + 0.3 * x
w_out = torch.randn(shape[0], width * 32, generator=g)
ref.train(); ref2.train(); net.train()
e32 = ref(x); (e32 * w_out).sum().backward()
em = forward_matched(ref2, x); (em * w_out).sum().backward()
emb = net(x.cuda()); (emb * w_out.cuda()).sum().backward()
print("emb: hip vs matched %.3e | hip vs fp32 %.3e | matched vs fp32 %.3e" % (rel(emb.detach().cpu(), em.detach()), rel(emb.detach().cpu(), e32.detach()), rel(em.detach(), e32.detach())))
p32, pm = dict(ref.named_parameters()), dict(ref2.named_parameters())
rows = []
for n, p in net.named_parameters():
    if n == "bn1.bias": continue
    rows.append((n, rel(p.grad.cpu(), pm[n].grad), rel(p.grad.cpu(), p32[n].grad), rel(pm[n].grad, p32[n].grad)))
for r in rows[:8] + rows[-8:]:
    print("%-32s hip-vs-matched %.3e  hip-vs-fp32 %.3e  matched-vs-fp32 %.3e" % r)
a = np.array([[r[1], r[2], r[3]] for r in rows])
print("grad rel-L2 median: hip-vs-matched %.3e hip-vs-fp32 %.3e matched-vs-fp32 %.3e" % tuple(np.median(a, 0)))
print("grad rel-L2 max   : hip-vs-matched %.3e hip-vs-fp32 %.3e matched-vs-fp32 %.3e" % tuple(np.max(a, 0)))
print("worst hip-vs-matched:", rows[int(np.argmax(a[:, 0]))])
