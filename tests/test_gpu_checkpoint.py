"""GPU: checkpoint compatibility with the reference's ``.h5`` files (mainKIT.py:169-170: ``torch.save(model.state_dict(),
path)`` of a DataParallel-wrapped net => torch-pickle, ``module.``-prefixed torchvision keys, contiguous OIHW weights)."""
import numpy as np
import pytest
import torch

from oracle.resnet50_reid import ResNet50ReID as OracleNet

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Encoders():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import Encoders
    return Encoders


def _reference_style_checkpoint(path, seed):
    torch.manual_seed(seed)
    ref = OracleNet(layers=(1, 1, 1, 1), width=32)
    ref.train()
    with torch.no_grad():
        for _ in range(2):                                   # non-trivial running statistics / num_batches_tracked
            ref(torch.randn(4, 3, 64, 32))
    sd = {"module." + k: v for k, v in ref.state_dict().items()}          # what DataParallel.state_dict() yields
    torch.save(sd, path)
    return ref.eval()


def test_load_reference_checkpoint(Encoders, tmp_path):
    path = str(tmp_path / "model_online_resnet50_Market.h5")
    ref = _reference_style_checkpoint(path, 3)
    online = Encoders._DataParallelShim(Encoders.ResNet50ReID(layers=(1, 1, 1, 1), width=32, seed=99))
    missing = online.load_state_dict(torch.load(path))                   # mainKIT-style load, strict
    assert not missing.missing_keys and not missing.unexpected_keys
    assert int(online.module.bn1.num_batches_tracked) == 2
    x = torch.randn(5, 3, 64, 32, generator=torch.Generator().manual_seed(1))
    online.eval()
    with torch.no_grad():
        y, y_ref = online(x.cuda()).cpu(), ref(x)
    rel = (y - y_ref).norm() / y_ref.norm()
    assert rel < 2e-2, float(rel)                                          # bf16 trunk vs fp32, running statistics
    # keys without the prefix load into the bare module too
    bare = Encoders.ResNet50ReID(layers=(1, 1, 1, 1), width=32, seed=5)
    bare.load_state_dict({k[len("module."):]: v for k, v in torch.load(path).items()})
    with torch.no_grad():
        assert torch.equal(bare.eval()(x.cuda()).cpu(), y)


def test_saved_checkpoint_loads_into_reference_layout(Encoders, tmp_path):
    online = Encoders._DataParallelShim(Encoders.ResNet50ReID(layers=(1, 1, 1, 1), width=32, seed=7))
    online.train()
    with torch.no_grad():
        online.module._run_forward(torch.randn(6, 3, 64, 32, device="cuda"), training=True)     # moves the BN statistics
    path = str(tmp_path / "model_online.h5")
    torch.save(online.state_dict(), path)                                  # mainKIT.py:169-170
    sd = torch.load(path, map_location="cpu")
    ref = OracleNet(layers=(1, 1, 1, 1), width=32)
    assert list(sd.keys()) == ["module." + k for k in ref.state_dict().keys()]                 # same names, same order
    for k, v in ref.state_dict().items():
        assert tuple(sd["module." + k].shape) == tuple(v.shape) and sd["module." + k].dtype == v.dtype, k
    ref.load_state_dict({k[len("module."):]: v for k, v in sd.items()})    # strict
    assert int(ref.bn1.num_batches_tracked) == 1
    x = torch.randn(5, 3, 64, 32, generator=torch.Generator().manual_seed(2))
    online.eval(); ref.eval()
    with torch.no_grad():
        y, y_ref = online(x.cuda()).cpu(), ref(x)
    assert (y - y_ref).norm() / y_ref.norm() < 2e-2
    # round trip through the file is exact
    again = Encoders._DataParallelShim(Encoders.ResNet50ReID(layers=(1, 1, 1, 1), width=32, seed=1))
    again.load_state_dict(torch.load(path))
    assert torch.equal(again.module.flat_params, online.module.flat_params)
    assert torch.equal(again.module.flat_buffers, online.module.flat_buffers)
    with torch.no_grad():
        assert torch.equal(again.eval()(x.cuda()).cpu(), y)
