"""Losses and regularisers of the adversarial step, with the reference's class names
(multi_stylegan/loss.py:97-170, 283-317, 353-395).  All of them return device tensors; nothing here forces a
host synchronisation."""
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import autograd


def _weighted(values: torch.Tensor, weight: Optional[torch.Tensor]) -> torch.Tensor:
    if weight is None:
        return values.mean()
    return (values * weight.view(1, 1, 1, weight.shape[-2], weight.shape[-1]).to(values.device)).mean()


class NonSaturatingLogisticGeneratorLoss(nn.Module):
    def forward(self, prediction_fake: torch.Tensor, weight: torch.Tensor = None) -> torch.Tensor:
        return _weighted(F.softplus(-prediction_fake), weight)


class NonSaturatingLogisticDiscriminatorLoss(nn.Module):
    def forward(self, prediction_real: torch.Tensor, prediction_fake: torch.Tensor,
                weight: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        return _weighted(F.softplus(-prediction_real), weight), _weighted(F.softplus(prediction_fake), weight)


class NonSaturatingLogisticDiscriminatorLossCutMix(nn.Module):
    """Per-pixel logistic loss against a binary CutMix label map (reference loss.py:173-196): the real term counts
    where the label is 1, the fake term where it is 0; both are means over ALL pixels."""

    def forward(self, prediction: torch.Tensor, label: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return (F.softplus(-prediction) * label).mean(), (F.softplus(prediction) * (1. - label)).mean()


class R1Regularization(nn.Module):
    def forward(self, prediction_real: torch.Tensor, image_real: torch.Tensor,
                prediction_real_pixel_wise: Optional[torch.Tensor] = None) -> torch.Tensor:
        outputs = prediction_real.sum() if prediction_real_pixel_wise is None else \
            (prediction_real.sum(), prediction_real_pixel_wise.sum())
        grad_real, = autograd.grad(outputs=outputs, inputs=image_real, create_graph=True)
        return 0.5 * grad_real.pow(2).reshape(grad_real.shape[0], -1).sum(1).mean()


class PathLengthRegularization(nn.Module):
    """Running-mean path-length penalty.  ``mean_path_length`` is a plain attribute in the reference (so it is lost
    on checkpoint, loss.py:369); here it is a persistent buffer and is all-reduced by the trainer under DDP."""

    def __init__(self, decay: float = 0.01) -> None:
        super().__init__()
        self.decay = decay
        self.register_buffer("mean_path_length", torch.zeros(1, dtype=torch.float))

    def forward(self, grad: torch.Tensor, reduce_fn=None) -> Tuple[torch.Tensor, torch.Tensor]:
        path_lengths = torch.sqrt(grad.pow(2).sum(2).mean(1) + 1e-08).mean()
        if reduce_fn is not None:
            # value of the global batch (the reference sees the gathered batch), gradient of the local shard:
            # averaged over ranks by the gradient all-reduce this reproduces the global-batch gradient exactly
            path_lengths = path_lengths + (reduce_fn(path_lengths.detach()) - path_lengths.detach())
        mean = self.mean_path_length.detach().to(grad.device)
        # NOT detached from path_lengths: the reference lets the gradient flow through the running mean
        # (loss.py:389-394), which scales the penalty gradient by (1 - decay)
        mean = mean + self.decay * (path_lengths - mean)
        penalty = torch.mean((path_lengths - mean) ** 2)
        self.mean_path_length = mean.detach()
        return penalty, path_lengths


class TopK(nn.Module):
    """Top-k training of the generator (reference loss.py:398-444): only the k = max(1, int(B v)) samples the
    discriminator rates most realistic contribute; v anneals linearly from 1 to 0.5 between `starting_iteration` and
    `final_iteration` (counted in forward calls).  Returns torch.topk's (values, indices) of the flattened scores."""

    def __init__(self, starting_iteration: int, final_iteration: int) -> None:
        super().__init__()
        self.starting_iteration = starting_iteration
        self.final_iteration = final_iteration
        self.iterations = 0

    def calc_v(self) -> float:
        self.iterations += 1
        if self.iterations <= self.starting_iteration:
            return 1.
        if self.iterations >= self.final_iteration:
            return 0.5
        progress = float(self.iterations - self.starting_iteration) / float(self.final_iteration - self.starting_iteration)
        return 0.5 * (1. - progress) + 0.5

    def forward(self, input: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        v = self.calc_v()
        scores = input.reshape(-1)
        return torch.topk(scores, k=max(1, int(scores.shape[0] * v)))
