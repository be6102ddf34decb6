"""Placeholders: the reference only touches these inside prepare_image(), which goldens do not exercise."""


class _Unavailable:
    def __getattr__(self, name):
        raise RuntimeError("stand-in torchvision.transforms has no implementation")


functional = _Unavailable()


def ToTensor():
    raise RuntimeError("stand-in torchvision.transforms has no implementation")
