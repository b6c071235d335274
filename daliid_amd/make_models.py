"""Mirror of the reference's ``make_models.py`` for the path in scope: ``make_model(cfg, num_class, camera_num, view_num)``
(make_models.py:399-410) -> ``build_transformer`` (:121-218): TransReID ViT encoder + BatchNorm1d neck with frozen bias,
``forward(x, label=None, cam_label=None, view_label=None)`` returning the post-neck ``feat`` (:184-205).

``cfg`` is the same yacs-like attribute tree the reference reads (cfg.MODEL.*, cfg.TEST.NECK_FEAT, cfg.INPUT.SIZE_TRAIN).
Out of scope: ``Backbone`` (broken in the reference, make_models.py:63) and ``build_transformer_local`` / JPM.
"""
from .vit_pytorch import ViTNeckNet

_GEOM = {  # factory name -> (embed_dim, depth, heads, mlp_ratio)   (vit_pytorch.py:453-476)
    "vit_base_patch16_224_TransReID": (768, 12, 12, 4.0),
    "deit_base_patch16_224_TransReID": (768, 12, 12, 4.0),
}


class build_transformer(ViTNeckNet):
    def __init__(self, num_classes, camera_num, view_num, cfg, factory=None, device=None, seed=None):
        name = cfg.MODEL.TRANSFORMER_TYPE
        if name not in _GEOM:
            raise NotImplementedError("build_transformer: transformer type %r is out of scope (head_dim 64 ViT-B only)" % name)
        if (cfg.MODEL.SIE_CAMERA and camera_num > 1) or (cfg.MODEL.SIE_VIEW and view_num > 1):
            raise NotImplementedError("build_transformer: SIE camera/view embeddings are out of scope")
        if cfg.MODEL.DROP_OUT != 0.0 or cfg.MODEL.ATT_DROP_RATE != 0.0:
            raise NotImplementedError("build_transformer: dropout is not supported")
        dim, depth, heads, ratio = _GEOM[name]
        print('using Transformer_type: {} as a backbone'.format(name))
        super().__init__(img_size=cfg.INPUT.SIZE_TRAIN, patch_size=16, stride_size=cfg.MODEL.STRIDE_SIZE, embed_dim=dim, depth=depth,
                         num_heads=heads, mlp_ratio=ratio, num_classes=1000, drop_path_rate=cfg.MODEL.DROP_PATH, device=device, seed=seed)
        self.neck, self.neck_feat, self.cos_layer = cfg.MODEL.NECK, cfg.TEST.NECK_FEAT, cfg.MODEL.COS_LAYER
        self.num_classes, self.ID_LOSS_TYPE = num_classes, cfg.MODEL.ID_LOSS_TYPE
        if cfg.MODEL.PRETRAIN_CHOICE == 'imagenet':
            raise NotImplementedError("build_transformer: ImageNet checkpoint loading needs a file fetched from the network; "
                                      "load a state_dict with load_state_dict instead")


def make_model(cfg, num_class, camera_num, view_num, device=None, seed=None):
    """make_models.py:399-410."""
    if cfg.MODEL.NAME != 'transformer':
        raise NotImplementedError("make_model: the ResNet `Backbone` branch is broken in the reference (make_models.py:63) and out of scope; "
                                  "use Encoders.getDCNN('resnet50')")
    if cfg.MODEL.JPM:
        raise NotImplementedError("make_model: build_transformer_local / JPM is out of scope (SURVEY 2.1)")
    model = build_transformer(num_class, camera_num, view_num, cfg, None, device=device, seed=seed)
    print('===========building transformer===========')
    return model
