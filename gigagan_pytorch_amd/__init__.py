"""gigagan_pytorch_amd — MI355X-native GigaGAN generator + discriminator training step.

Same public names as the reference package (gigagan_pytorch/__init__.py:1-31).
"""
from .version import __version__
from .modules import AdaptiveConv2DMod, StyleNetwork
from .text import TextEncoder
from .generator import Generator
from .discriminator import Discriminator
from .gigagan import GigaGAN
from .data import ImageDataset, TextImageDataset, MockTextImageDataset


def __getattr__(name):
    if name == 'UnetUpsampler':
        from .unet_upsampler import UnetUpsampler
        return UnetUpsampler
    if name == 'VisionAidedDiscriminator':
        raise AttributeError('VisionAidedDiscriminator is out of scope for the MI355X build (needs a CLIP vision tower)')
    raise AttributeError(name)


__all__ = ['GigaGAN', 'Generator', 'Discriminator', 'AdaptiveConv2DMod', 'StyleNetwork', 'UnetUpsampler',
           'TextEncoder', 'ImageDataset', 'TextImageDataset', 'MockTextImageDataset']
