"""GPU: mainKIT.main's call sequence (mainKIT.py:73-176) replayed line for line against the mirrors, on SyntheticImages.

The reference driver itself cannot be imported (it needs torchvision / torchreid / termcolor and hard-coded dataset paths,
SURVEY.md 8c), so the sequence is restated here with its line numbers: getDCNN :73 -> getValidator / setParameters / validate :85-87 ->
torch.optim.Adam(model_online.parameters()) :99 -> trainer(...) :118-120 -> LR table :129-132 -> per epoch lambda_lr_warmup :144, :204-208
-> train :148 -> validate online + momentum :158-163 -> torch.save of both state_dicts :169-170.  Only the dataset loader is replaced
(load_dataset's /scratch paths are out of scope); every other object is what an unchanged mainKIT.py would get after the import swap
of INTEGRATION.md."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def lambda_lr_warmup(optimizer, lr_value, weight_decay_value):            # mainKIT.py:204-208
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr_value
        param_group['weight_decay'] = weight_decay_value


def test_mainkit_call_sequence(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd.Encoders import getDCNN, getEnsembles                 # mainKIT.py:27
    from daliid_amd.validateModels import validationManager, MSMT17_validator   # :28
    from daliid_amd.train_encodersKIT import trainer                      # :30
    from daliid_amd.getFeatures import extractFeatures, get_subset_one_encoder  # :39
    from daliid_amd import synthetic
    assert callable(getEnsembles) and callable(get_subset_one_encoder) and callable(extractFeatures) and MSMT17_validator is not None
    np.random.seed(12); torch.manual_seed(12)                             # :47-48
    data = synthetic.SyntheticImages(n_ids=12, per_id=6, n_cams=3, seed=5, noise=0.4).install()
    try:
        img_height, img_width, model_name, base_lr, weight_decay = 64, 32, "resnet50", 3.5e-4, 5e-4
        P, K, tau, beta, lambda_proxy, num_iter, number_of_epoches, eval_freq, version = 4, 4, 0.05, 0.999, 0.4, 1, 2, 1, "v1"
        dataset, dir_to_save = "Synthetic", str(tmp_path)
        gpu_indexes = [0]
        model_online, model_momentum = getDCNN(gpu_indexes, model_name)                                  # :73
        train_images_dataset, gallery_images_dataset, queries_images_dataset = data.split(1)             # :81 (load_dataset)
        validator = validationManager.getValidator(dataset)                                              # :85
        validator.setParameters(img_height, img_width, False, gpu_indexes[0])                            # :86
        cmc0, mAP0, _ = validator.validate(queries_images_dataset, gallery_images_dataset, model_online)  # :87
        optimizer = torch.optim.Adam(model_online.parameters(), lr=base_lr, weight_decay=weight_decay)   # :99
        selected_images = train_images_dataset
        selected_labels = np.int32(train_images_dataset[:, 1])
        labels = np.unique(selected_labels)
        labels_dict = {labels[idx]: idx for idx in np.arange(len(labels))}
        model_trainer = trainer(dataset, selected_images, model_name, labels_dict, img_height, img_width, None, False, 0, optimizer, P, K, tau,
                                beta, lambda_proxy, number_of_epoches, model_online, model_momentum, gpu_indexes, version)   # :118-120
        base_lr_values = np.concatenate((np.linspace(base_lr, base_lr, num=100), np.linspace(base_lr / 10, base_lr / 10, num=100),
                                         np.linspace(base_lr / 100, base_lr / 100, num=50)))             # :129-132
        base_weight_decay_value = np.linspace(weight_decay, weight_decay, num=number_of_epoches)         # :134
        w_start = model_online.module.flat_params.clone()
        best_cmc, saved, losses = 0, [], []
        for pipeline_iter in range(1, number_of_epoches + 1):
            lambda_lr_warmup(model_trainer.optimizer, base_lr_values[pipeline_iter - 1], base_weight_decay_value[pipeline_iter - 1])   # :144
            model_trainer.train(selected_images, selected_labels, num_iter, pipeline_iter)               # :148
            losses.append(model_trainer.last_epoch_stats["loss"])
            assert model_trainer.last_epoch_stats["steps"] >= 1
            if pipeline_iter % eval_freq == 0:
                cmc, mAP, _ = validator.validate(queries_images_dataset, gallery_images_dataset, model_online)          # :158
                validator.validate(queries_images_dataset, gallery_images_dataset, model_momentum)                      # :159
                if cmc[0] > best_cmc or not saved:                        # (:161 saves on improvement; the first evaluation always improves on 0
                    best_cmc = cmc[0]                                     #  unless rank-1 is exactly 0: keep the two torch.save calls covered)
                    for tag, m in (("online", model_trainer.model_online), ("momentum", model_trainer.model_momentum)):
                        path = "%s/model_%s_%s_%s.h5" % (dir_to_save, tag, model_name, version)          # :169-170
                        torch.save(m.state_dict(), path)
                        saved.append((path, m.module.flat_params.clone()))
                assert np.isfinite(mAP) and 0.0 <= mAP <= 1.0 and len(cmc) > 0
        # the optimizer mainKIT built is the one that was stepped: the weights moved by ~lr per step, the momentum model by (1 - beta) of it
        moved = (model_online.module.flat_params - w_start).abs().max().item()
        assert 1e-4 < moved < 1e-2, moved
        d_mom = (model_momentum.module.flat_params - w_start).abs().max().item()
        assert 0 < d_mom < moved
        assert all(np.isfinite(l) for l in losses)
        # the checkpoints are the reference's format: torch-pickled state_dict, torchvision keys under `module.`, and load back bit for bit
        assert len(saved) == 2
        sd = torch.load(saved[0][0], map_location="cpu")
        assert all(k.startswith("module.") for k in sd) and "module.layer4.2.conv3.weight" in sd and "module.last_bn.running_var" in sd
        on2, _ = getDCNN(gpu_indexes, model_name)
        on2.load_state_dict(sd)
        assert torch.equal(on2.module.flat_params, saved[0][1])             # the weights as they were when the checkpoint was written
        # lambda_lr_warmup's values are what the fused step reads (mainKIT.py:144 -> param_groups)
        lambda_lr_warmup(model_trainer.optimizer, base_lr_values[150], 1e-4)
        assert model_trainer._adam.hyper()[0] == pytest.approx(base_lr / 10) and model_trainer._adam.hyper()[3] == pytest.approx(1e-4)
    finally:
        synthetic.SyntheticImages.uninstall()


def test_resnet50reid_takes_a_torchvision_shaped_model_base():
    """Encoders.py:33-37: ResNet50ReID(resnet50(pretrained=True)).  torchvision is absent here, so the stand-in is a module with exactly
    torchvision's resnet50 state_dict keys: the trunk + `fc.*`, no `last_bn.*` (oracle/resnet50_reid.py builds the same trunk)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from daliid_amd import Encoders
    from daliid_amd._lib import DaliError
    src = Encoders.ResNet50ReID(seed=3)
    sd = {k: v.detach().cpu().clone() for k, v in src.state_dict().items() if not k.startswith("last_bn.")}
    sd["fc.weight"], sd["fc.bias"] = torch.randn(1000, 2048), torch.randn(1000)

    class Base(torch.nn.Module):
        def state_dict(self, *a, **k):
            return sd

    net = Encoders.ResNet50ReID(Base())
    got = net.state_dict()
    for k, v in sd.items():
        if not k.startswith("fc."):
            assert torch.equal(got[k].cpu(), v), k
    assert not any(k.startswith("fc.") for k in got)
    assert torch.equal(got["last_bn.weight"].cpu(), torch.ones(2048)) and torch.equal(got["last_bn.running_var"].cpu(), torch.ones(2048))
    bad = dict(sd); bad.pop("layer3.0.conv1.weight")
    with pytest.raises(DaliError):
        Encoders.ResNet50ReID(bad)
