"""Layer-by-layer comparison of the HIP plan against the fp32 oracle (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.resnet50_reid import ResNet50ReID as OracleNet
from daliid_amd import Encoders
bf16 = torch.bfloat16

def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))

if len(sys.argv) > 1 and sys.argv[1] == "full":
    layers, width, shape = (3, 4, 6, 3), 64, (32, 3, 256, 128)
else:
    layers, width, shape = (2, 1, 2, 1), 64, (5, 3, 96, 48)
torch.manual_seed(2)
ref = OracleNet(layers=layers, width=width)
gam = float(os.environ.get("BN3_GAMMA", "1.0"))
with torch.no_grad():
    for m in ref.modules():
        if hasattr(m, "bn3"):
            m.bn3.weight.mul_(gam)
net = Encoders.ResNet50ReID(layers=layers, width=width)
net.load_state_dict(ref.state_dict())
x = torch.randn(*shape, generator=torch.Generator().manual_seed(12))
acts = {}
def hook(name):
    def f(m, i, o): acts[name] = o.detach()
    return f
blocks = [b for l in (ref.layer1, ref.layer2, ref.layer3, ref.layer4) for b in l]
for i, b in enumerate(blocks):
    b.conv1.register_forward_hook(hook("block%d.raw1" % i)); b.conv2.register_forward_hook(hook("block%d.raw2" % i))
    b.conv3.register_forward_hook(hook("block%d.raw3" % i)); b.register_forward_hook(hook("block%d.y" % i))
    if b.downsample is not None: b.downsample[0].register_forward_hook(hook("block%d.rawd" % i))
ref.maxpool.register_forward_hook(hook("pool0"))
ref.train(); net.train()
emb_ref = ref(x)
with torch.no_grad():
    emb = net(x.cuda())
for name, t in acts.items():
    n, c, h, w = t.shape
    got = net.debug_tensor(name, bf16, (n, h, w, c)).float().cpu().permute(0, 3, 1, 2)
    print("%-14s shape %-18s rel-L2 %.3e" % (name, tuple(t.shape), rel(got, t)))
print("emb rel", rel(emb.cpu(), emb_ref.detach()))
