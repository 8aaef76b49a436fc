"""gigagan_pytorch_amd — MI355X-native GigaGAN generator + discriminator training step."""
__version__ = '0.1.0'
