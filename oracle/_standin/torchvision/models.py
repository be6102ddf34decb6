"""`vgg16()` with torchvision's VGG-16 "D" `features` layout and injected weights.

The layout (conv3x3+ReLU runs separated by MaxPool2d, 31 modules) is the public
VGG-16 configuration D of Simonyan & Zisserman; the reference slices it by index
(DISTS_pt.py:36-49), so only the module order matters.  There is no download
path: weights come from `WEIGHT_PROVIDER()` (set by make_goldens.py), a list of 13
(weight, bias) numpy pairs.
"""
import torch
import torch.nn as nn

WEIGHT_PROVIDER = None
_CFG_D = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")


class _VGG(nn.Module):
    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in _CFG_D:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)


def vgg16(pretrained=False, **kwargs):
    net = _VGG()
    if WEIGHT_PROVIDER is None:
        raise RuntimeError("stand-in torchvision: set models.WEIGHT_PROVIDER first (no download path)")
    convs = [m for m in net.features if isinstance(m, nn.Conv2d)]
    for m, (w, b) in zip(convs, WEIGHT_PROVIDER()):
        m.weight.data = torch.from_numpy(w).clone()
        m.bias.data = torch.from_numpy(b).clone()
    return net
