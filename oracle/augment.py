"""Oracle: the reference's per-image transforms on PIL images, restated with PIL itself + numpy.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  torchvision is absent from the image; what it does for these
transforms on a PIL image is: ``Resize`` -> ``img.resize((w, h), BICUBIC)``; ``RandomCrop(padding)`` -> zero pad + crop;
``RandomHorizontalFlip`` -> transpose(FLIP_LEFT_RIGHT); ``ColorJitter`` -> ``ImageEnhance.{Brightness, Contrast, Color}``
in the drawn order; ``ToTensor`` -> /255 fp32 CHW; ``RandomErasing`` -> zero box; ``Normalize`` -> (x - mean) / std
(getFeatures.py:18-19, train_encodersKIT.py:313-320).  PIL (12.2 in this image) is the pin for the pixel arithmetic."""
import numpy as np
from PIL import Image, ImageEnhance

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(3, 1, 1)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(3, 1, 1)


def resize(img_u8, out_h, out_w):
    return np.asarray(Image.fromarray(img_u8).resize((out_w, out_h), Image.BICUBIC))


def to_tensor_normalize(img_u8):
    x = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    return (x - MEAN) / STD


def train_transform(img_u8, p):
    """img_u8: resized uint8 [H,W,3]; p: one row of daliid_amd.transforms.sample_train_params -> fp32 [3,H,W]."""
    H, W, _ = img_u8.shape
    top, left, flip = int(p[0]), int(p[1]), int(p[2])
    order = [int(v) for v in p[3:7]]
    ei, ej, eh, ew = (int(v) for v in p[7:11])
    factors = np.array(p[11:14], dtype=np.int32).view(np.float32)
    pad = int(p[14])
    padded = np.zeros((H + 2 * pad, W + 2 * pad, 3), dtype=np.uint8)
    padded[pad:pad + H, pad:pad + W] = img_u8
    img = Image.fromarray(padded[top:top + H, left:left + W])
    if flip:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    for op in order:
        if op == 0:
            img = ImageEnhance.Brightness(img).enhance(float(factors[0]))
        elif op == 1:
            img = ImageEnhance.Contrast(img).enhance(float(factors[1]))
        elif op == 2:
            img = ImageEnhance.Color(img).enhance(float(factors[2]))
    x = np.asarray(img).astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    if eh > 0:
        x[:, ei:ei + eh, ej:ej + ew] = 0.0
    return (x - MEAN) / STD
