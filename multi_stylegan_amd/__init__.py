"""Multi-StyleGAN generator + discriminator training hot path, MI355X-native (gfx950 HIP kernels behind the
reference's nn.Module / op_static API).  See DESIGN.md.  Public names follow multi_stylegan/__init__.py."""
from .adaptive_discriminator_augmentation import AdaptiveDiscriminatorAugmentation, AugmentationPipeline
from .config import (generation_hyperparameters, multi_style_gan_generator_config,
                     u_net_2d_discriminator_config)
from .data import DevicePrefetcher, SyntheticBatches
from .inference import GeneratorSampler, load_generator_ema, split_sequences, validation_samples
from .loss import PathLengthRegularization, TopK
from .model_wrapper import Draws, ModelWrapper
from .multi_stylegan_generator import Generator as MultiStyleGANGenerator
from .u_net_2d_discriminator import Discriminator as MultiStyleGANDiscriminator
from .validation_metrics import FID, FVD, IS

__all__ = ["MultiStyleGANGenerator", "MultiStyleGANDiscriminator", "ModelWrapper", "Draws", "PathLengthRegularization",
           "TopK", "AdaptiveDiscriminatorAugmentation", "AugmentationPipeline", "GeneratorSampler", "load_generator_ema", "split_sequences", "validation_samples",
           "DevicePrefetcher", "SyntheticBatches", "IS", "FID", "FVD", "multi_style_gan_generator_config", "u_net_2d_discriminator_config", "generation_hyperparameters"]
