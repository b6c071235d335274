"""Weight gradient of a 1x1 convolution under cancellation (the conv -> BatchNorm case: the raw-output gradient is orthogonal to 1 and to the
raw output per channel, so dW = sum_p dy[p] x[p]^T is a small residual of large terms): kernel vs fp64, at layer1's shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliid_amd import ops_nn as nn
bf16 = torch.bfloat16
N, H, W = 256, 64, 32
for cin, cout in ((64, 64), (256, 64), (256, 128)):
    g = torch.Generator(device="cuda").manual_seed(1)
    P = N * H * W
    x = torch.randn(P, cin, device="cuda", generator=g).to(bf16)
    noise = torch.randn(P, cout, device="cuda", generator=g)
    xd = x.double()
    G = xd.T @ xd
    coef = torch.linalg.solve(G, xd.T @ noise.double())                 # least squares: remove everything correlated with x
    S = torch.randn(cin, cout, device="cuda", generator=g).double() * 0.02
    dy = (noise.double() - xd @ coef + xd @ S).float().to(bf16)          # dW_true ~= G S (+ rounding of dy)
    ref = (dy.double().T @ xd)                                           # [cout, cin] exact for the bf16 operands
    got = nn.conv2d_wgrad(x.view(N, H, W, cin), dy.view(N, H, W, cout), (1, 1)).view(cout, cin).double()
    inc = ((dy.double() ** 2).T @ (xd ** 2)).sqrt()
    print("cin %d cout %d: |dW| / incoherent norm = %.3e ; kernel vs fp64 rel-L2 %.3e ; max-abs/max %.3e"
          % (cin, cout, float(ref.norm() / inc.norm()), float((got - ref).norm() / ref.norm()), float((got - ref).abs().max() / ref.abs().max())))
