"""Hyper-parameter dictionaries with the reference's names and values (multi_stylegan/config.py:6-57)."""
import math
from typing import Any, Dict

u_net_2d_discriminator_config: Dict[str, Any] = {
    "encoder_channels": ((3, 128), (128, 256), (256, 384), (384, 768), (768, 1024)),
    "decoder_channels": ((1024, 768), (768, 384), (384, 256), (256, 128)),
    "fft": False,
}

multi_style_gan_generator_config: Dict[str, Any] = {
    "channels": (512, 512, 512, 512, 512, 512, 512),
    "channel_factor": 1,
    "latent_dimensions": 512,
    "depth_style_mapping": 8,
    "starting_resolution": (4, 4),
}

generation_hyperparameters: Dict[str, Any] = {
    "p_mixed_noise": 0.9,
    "lazy_generator_regularization": 16,
    "w_generator_regularization": math.log(2) / ((256 ** 2) * (math.log(256) - math.log(2))),
    "lazy_discriminator_regularization": 16,
    "w_discriminator_regularization_r1": 10.0,
    "w_discriminator_regularization": 4.0,
    "batch_factor_wrong_order": 1. / 4.,
    "batch_size_shrink_path_length_regularization": 2. / 4.,
    "betas": (0.0, 0.999),
    "top_k_start": 1. / 4.,
    "top_k_finish": 3. / 4.,
    "wrong_order_start": 3. / 4.,
    "trap_weight": 1. / 4.,
}


def generator_config_for_resolution(resolution: int, width: int = 512) -> Dict[str, Any]:
    """BASELINE.json configs: 64 -> 5 stages, 256 -> 7 (the default), 512 -> 8 (resolution = 4 * 2^(stages-1))."""
    stages = int(round(math.log2(resolution // 4))) + 1
    assert 4 * 2 ** (stages - 1) == resolution, resolution
    cfg = dict(multi_style_gan_generator_config)
    cfg["channels"] = (width,) * stages
    return cfg
