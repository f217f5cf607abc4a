"""UNet definitions. The reference selects sd/sdxl at import time from env
``DIFFUSERS_REWRITE`` (diffusers_rewrite/__init__.py:1-6); here the class takes an
``arch`` argument and only *defaults* to that env var."""
from .unet import (ARCH, Timesteps, TimestepEmbedding, ResnetBlock2D, Attention, GEGLU, FeedForward,
                   BasicTransformerBlock, Transformer2DModel, Downsample2D, Upsample2D,
                   UNet2DConditionModel)
