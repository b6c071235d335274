"""GPU: the weight gradients on the plan's second stream (DALI_WGRAD_STREAM=1, resnet_plan.hip wgrad_fork / before_write / wgrad_join) must leave
the flat gradient buffer BIT-IDENTICAL to the single-stream order: the same kernels with the same summation order run, only on two streams, so
any difference is a missing dependency (a gradient buffer overwritten while a weight-gradient GEMM still reads it)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _grads(net, x, d_emb, mode, reps):
    from daliid_amd import _lib
    os.environ["DALI_WGRAD_STREAM"] = str(mode)
    _lib.lib().dali_debug_reload_env()
    out = []
    for _ in range(reps):
        net.flat_grads.fill_(float("nan"))
        emb = net._run_forward(x, training=True)
        net._run_backward(d_emb)
        torch.cuda.synchronize()
        out.append((emb.clone(), net.flat_grads.clone()))
    return out


@pytest.mark.parametrize("batch,h,w", [(32, 128, 64), (64, 256, 128)])
def test_side_stream_gradients_bit_identical(batch, h, w):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import Encoders, _lib
    dev = torch.device("cuda", 0)
    net = Encoders.ResNet50ReID(device=dev, seed=5)
    net.train()
    gen = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(batch, 3, h, w, device=dev, generator=gen)
    d_emb = torch.randn(batch, 2048, device=dev, generator=gen) * 1e-2
    try:
        ref = _grads(net, x, d_emb, 0, 1)[0]
        assert torch.isfinite(ref[1]).all()
        for mode in (1, 2):                                        # 1: beside everything, 2: beside the BatchNorm passes only
            for emb, g in _grads(net, x, d_emb, mode, 4):          # several rounds: the six gradient buffers rotate, stale readers would show
                assert torch.equal(emb, ref[0])
                assert torch.equal(g, ref[1]), (mode, float((g - ref[1]).abs().max()))
        back = _grads(net, x, d_emb, 0, 1)[0]
        assert torch.equal(back[1], ref[1])
    finally:
        os.environ.pop("DALI_WGRAD_STREAM", None)
        _lib.lib().dali_debug_reload_env()
