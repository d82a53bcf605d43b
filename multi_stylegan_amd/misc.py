"""Hot-path helpers with the reference's names: style-mixing noise, the EMA step and the batch-wise range
normalisations of the validation metrics (multi_stylegan/misc.py:183-199, 216-235, 238-252)."""
import random
from typing import List, Union

import torch


def get_noise(batch_size: int, latent_dimension, p_mixed_noise: float = 0.9, device: str = "cuda") -> Union[
        torch.Tensor, List]:
    if (p_mixed_noise > 0) and (random.random() < p_mixed_noise):
        return list(torch.randn(2, batch_size, latent_dimension, dtype=torch.float32, device=device).unbind(0))
    return torch.randn(batch_size, latent_dimension, dtype=torch.float32, device=device)


def normalize_0_1_batch(input: torch.Tensor) -> torch.Tensor:
    """Every sample of a 5-D batch mapped to its own [0, 1] range and then clamped from BELOW at 1e-3 (reference
    misc.py:216-225: the clamp is the reference's, values under 1e-3 are raised to it)."""
    flat = input.reshape(input.shape[0], -1)
    lo = flat.min(dim=1)[0][:, None, None, None, None]
    hi = flat.max(dim=1)[0][:, None, None, None, None]
    return ((input - lo) / (hi - lo)).clamp(min=1e-03)


def normalize_m1_1_batch(input: torch.Tensor) -> torch.Tensor:
    """2 * normalize_0_1_batch(x) - 1 (reference misc.py:228-235)."""
    return 2. * normalize_0_1_batch(input) - 1.


def random_permutation(n: int) -> torch.Tensor:
    """Index vector for the wrongly-ordered-sequence augmentation (reference misc.py:202-213).  As in the reference the
    n indices are drawn WITH replacement from numpy's global generator -- a time step may repeat -- and only the
    identity is excluded: it is replaced by the reversed order."""
    import numpy as np
    permutation = torch.from_numpy(np.random.choice(range(n), size=n))
    if torch.equal(permutation, torch.arange(n)):
        permutation = torch.arange(n - 1, -1, -1)
    return permutation


@torch.no_grad()
def exponential_moving_average(model_ema, model_train, decay: float = 0.999) -> None:
    """ema <- decay * ema + (1 - decay) * train over the named parameters, as ONE multi-tensor launch."""
    assert type(model_ema) is type(model_train), "EMA can only be performed on networks of the same type!"
    train = dict(model_train.named_parameters())
    ema_params, src = [], []
    for name, p in model_ema.named_parameters():
        ema_params.append(p.data)
        src.append(train[name].data)
    torch._foreach_mul_(ema_params, decay)
    torch._foreach_add_(ema_params, src, alpha=1 - decay)
    from . import conv_ops
    # `.data` writes do not bump the tensors' version counters: declare the EMA copy's parameters (only those) changed
    conv_ops.invalidate_weight_cache(list(model_ema.parameters()))
