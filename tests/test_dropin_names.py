"""CPU: the names an unchanged mainKIT.py / train_encodersKIT.py / evaluateCleanATModels.py import from the modules this package mirrors
exist under the same module names (SURVEY.md 8b "signatures to keep").  The lists are data, restated from the reference's import
lines; out-of-scope names are stubs that raise NotImplementedError when CALLED, never at import."""
import importlib
import inspect

import pytest

# (mirror module, names, reference import line)
IMPORTS = [
    ("Encoders", ["getDCNN", "getEnsembles"], "mainKIT.py:27, train_encodersKIT.py:25, evaluateCleanATModels.py:13"),
    ("validateModels", ["validationManager", "MSMT17_validator"], "mainKIT.py:28"),
    ("train_encodersKIT", ["trainer"], "mainKIT.py:30"),
    ("getFeatures", ["extractFeatures", "get_subset_one_encoder"], "mainKIT.py:39"),
    ("losses", ["BatchWeightedCenterLoss", "BatchWeightedProxyLoss", "getValueFromCosineSchedule", "getACCBal"], "train_encodersKIT.py:27 (from losses import *)"),
    ("make_models", ["make_model"], "Encoders.py:20"),
    ("vit_pytorch", ["vit_base_patch16_224_TransReID"], "make_models.py:5"),
]
OUT_OF_SCOPE_STUBS = [("Encoders", "getEnsembles", ([0],)), ("getFeatures", "get_subset_one_encoder", (None, None, 5, None)),
                      ("validateModels", "MSMT17_validator", (None, None, None, "."))]


@pytest.mark.parametrize("module,names,where", IMPORTS)
def test_reference_import_lines_resolve(module, names, where):
    m = importlib.import_module("daliid_amd." + module)
    for n in names:
        assert hasattr(m, n), "%s.%s (imported at %s) is missing" % (module, n, where)


@pytest.mark.parametrize("module,name,args", OUT_OF_SCOPE_STUBS)
def test_out_of_scope_names_fail_loudly_when_called(module, name, args):
    fn = getattr(importlib.import_module("daliid_amd." + module), name)
    with pytest.raises(NotImplementedError, match="out of scope"):
        fn(*args)


def test_signatures_kept():
    from daliid_amd import Encoders, getFeatures, train_encodersKIT, validateModels
    assert list(inspect.signature(Encoders.getDCNN).parameters) == ["gpu_indexes", "model_name", "embedding_size"]
    assert list(inspect.signature(Encoders.getEnsembles).parameters) == ["gpu_indexes"]
    assert list(inspect.signature(Encoders.ResNet50ReID.__init__).parameters)[:2] == ["self", "model_base"]
    assert list(inspect.signature(getFeatures.get_subset_one_encoder).parameters) == ["selected_sample", "train_set", "topK", "encoder", "batch_size", "gpu_index"]
    assert list(inspect.signature(validateModels.MSMT17_validator.__init__).parameters) == ["self", "train_images", "val_images", "trainer", "dir_to_save"]
    assert list(inspect.signature(train_encodersKIT.trainer.__init__).parameters)[1:21] == [
        "dataset", "selected_images", "model_name", "labels_dict", "img_height", "img_width", "turbulance_dir_path", "is_clean_training",
        "kind_of_transform", "optimizer", "P", "K", "tau", "beta", "lambda_proxy", "number_of_epoches", "model_online", "model_momentum",
        "gpu_indexes", "version"]
    assert list(inspect.signature(train_encodersKIT.trainer.train).parameters) == ["self", "selected_images", "selected_labels", "number_of_iterations", "current_epoch"]
