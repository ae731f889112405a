"""MI355X-native VQ-VAE-2 stage-1 hot path (drop-in for the reference's vqvae.py).

Import as ``vqvae2_amd`` (see /vqvae2_amd.py: the directory name carries hyphens).
Public surface mirrors /root/reference/vqvae.py and distributed/__init__.py:1-13.
"""
from .vqvae import VQVAE, Quantize, ResBlock, Encoder, Decoder, Conv2d, ConvTranspose2d, ReLU  # noqa: F401
from . import vqvae_deep  # noqa: F401
from .vqvae_deep import VQVAE_Deep  # noqa: F401
from . import distributed  # noqa: F401
from . import ops  # noqa: F401
from . import codes  # noqa: F401
from .optim import FusedAdam, CycleScheduler  # noqa: F401
from .train import Stage1Trainer, stage1_loss  # noqa: F401
