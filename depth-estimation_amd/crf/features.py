"""``Vgg16features`` as the reference notebooks import it from ``crf.features`` (crf/features.py:9-67).
Out of the hot path: the 5-D bilateral features never use it (Experiments/DenseCrf.ipynb:146 is
commented out); it is here so that the notebook's cell 4 executes.

The reference constructs ``torchvision.models.vgg16(pretrained=True)``, i.e. a network download.
Offline (or without torchvision) the same 4-block VGG-16 convolution stack is built with a fixed
seed instead and a warning is issued: shapes and API are right, the feature VALUES are not the
ImageNet ones.  Parity unpinned: no reference output exists for this class in the tree."""
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401   (re-exported like the reference module does)
from scipy.ndimage import zoom

_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512]      # vgg16.features[:23]
_TAPS = {3, 8, 15, 22}                                                        # relu1_2, relu2_2, relu3_3, relu4_3


def _vgg_stack():
    try:
        import torchvision.models as models

        return nn.ModuleList(list(models.vgg16(pretrained=True).features)[:23]), True
    except Exception:  # no torchvision / no network
        layers, cin = [], 3
        g = torch.Generator().manual_seed(16)
        for v in _CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                conv = nn.Conv2d(cin, v, 3, padding=1)
                with torch.no_grad():
                    conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (9 * cin)) ** 0.5)
                    conv.bias.zero_()
                layers += [conv, nn.ReLU(inplace=False)]
                cin = v
        return nn.ModuleList(layers[:23]), False


class Vgg16features(nn.Module):
    def __init__(self):
        super().__init__()
        self.features, self.pretrained = _vgg_stack()
        self.features.eval()
        if not self.pretrained:
            warnings.warn("Vgg16features: pretrained VGG-16 weights are not available offline; using a "
                          "fixed-seed random stack of the same architecture (feature values are placeholders)")

    def forward(self, x):
        out = []
        for i, layer in enumerate(self.features):
            x = layer(x)
            if i in _TAPS:
                out.append(x)
        return out

    def preprocess(self, x):
        """[h, w, 3] numpy image in [0, 1] -> normalised (1, 3, 224, 224) tensor."""
        mean = np.array([0.29298669, 0.26512041, 0.21699697])
        std = np.array([0.24798678, 0.19988715, 0.18761264])
        t = torch.from_numpy(((x - mean) / std)[None].transpose(0, 3, 1, 2)).float()
        t = t.to(next(self.features.parameters()).device)
        return nn.functional.interpolate(t, size=(224, 224), mode="bilinear", align_corners=True)

    def rescale_reshape(self, img_torch, img_shape):
        h, w, _ = img_shape
        a = img_torch.detach().cpu().numpy()[0].transpose(1, 2, 0)
        return zoom(a, (h / a.shape[0], w / a.shape[1], 1), order=2)

    def get_all_features(self, x):
        with torch.no_grad():
            return [self.rescale_reshape(f, x.shape) for f in self(self.preprocess(x))]

    def get_features(self, x, k=3):
        return self.get_all_features(x)[:k]

    def get_torch_features(self, x, k=0):
        x = nn.functional.interpolate(x, size=(224, 224), mode="bilinear", align_corners=True)
        x = (x - x.mean(dim=1, keepdim=True)) / (x.std(dim=1, keepdim=True) + 1e-8)
        return self(x)[k].data

    def get_random_features(self, x, i=0, num_features=10):
        f = self.get_all_features(x)[i]
        p = f @ np.random.rand(f.shape[-1], num_features)
        return (p - p.mean((0, 1))) / p.std((0, 1))
