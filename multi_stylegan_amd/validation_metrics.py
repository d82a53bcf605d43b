"""Sample-quality statistics of the validation pass (reference multi_stylegan/validation_metrics.py, call site
model_wrapper.py:197-243): inception score, Frechet inception distance, Frechet video distance.

What is here is the part of that file that is arithmetic: the per-channel frame selection, the feature sweeps over the
dataset and the EMA generator, and the statistics on the features.  What is NOT here are the feature networks: the
reference instantiates torchvision's pretrained Inception-v3 and a pretrained I3D (validation_metrics.py:48, 571-650) whose
weights are missing blobs of the reference repository and cannot be downloaded -- every metric class therefore takes the
network as an argument (any ``nn.Module`` mapping a ``[B, 3, H, W]`` image / ``[B, 3, T, H, W]`` clip batch in [-1, 1] to
features or logits); with the pretrained networks supplied the classes compute the reference's numbers.

MI355X-first: the statistics stay on the GPU.  The reference moves every activation to the host, stacks 5000 x 2048
arrays and calls numpy / scipy.  Here a feature sweep folds each batch into running first and second moments (float64, one
``[d, d]`` GEMM per batch on the device), and tr(sqrtm(C_r C_f)) is taken as the sum of the square roots of the eigenvalues
of the SYMMETRIC matrix C_r^(1/2) C_f C_r^(1/2) (same spectrum as C_r C_f; ``eigh`` on the device, no complex arithmetic,
eigenvalues clamped at zero where scipy's ``sqrtm(...).real`` drops an imaginary part).
"""
import math
import warnings
from typing import Iterable, Optional, Tuple, Union

import torch
import torch.distributed as torch_dist
import torch.nn as nn

from . import misc


def _ranks() -> int:
    """Ranks that share a validation pass: in a data-parallel job every rank sweeps 1 / world of the samples (its own shard
    of the dataset, its own latents) and the additive statistics are summed over the ranks -- the reference runs the pass once,
    on the gathered DataParallel model (model_wrapper.py:197-243); repeating the whole pass per rank was world x the work."""
    return torch_dist.get_world_size() if torch_dist.is_available() and torch_dist.is_initialized() else 1

__all__ = ["IS", "FID", "FVD", "FeatureMoments", "frechet_distance", "frechet_distance_from_moments", "inception_score",
           "select_frames"]


def select_frames(images: torch.Tensor, channel: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """validation_metrics.py:93-94, 246-247: channel ``c`` of ONE random time step of the batch (one ``torch.randint`` draw
    from the global CPU generator, like the reference), replicated on three colour planes: [B, C, T, H, W] -> [B, 3, 1, H, W]."""
    t = torch.randint(0, images.shape[2], (1,), generator=generator)
    return images[:, channel, t.to(images.device)].unsqueeze(dim=1).repeat_interleave(dim=1, repeats=3)


def select_clips(images: torch.Tensor, channel: int) -> torch.Tensor:
    """validation_metrics.py:451-452: the whole sequence of channel ``c`` on three colour planes: -> [B, 3, T, H, W]."""
    return images[:, channel].unsqueeze(dim=1).repeat_interleave(dim=1, repeats=3)


class FeatureMoments:
    """Running sample count, sum and sum of outer products of feature rows, float64 on the features' device.  ``limit``
    rows at most are taken (the reference truncates its activation lists to ``data_samples``)."""

    def __init__(self, limit: Optional[int] = None):
        self.limit, self.n = limit, 0
        self.s1: Optional[torch.Tensor] = None
        self.s2: Optional[torch.Tensor] = None

    @property
    def full(self) -> bool:
        return self.limit is not None and self.n >= self.limit

    def update(self, features: torch.Tensor) -> "FeatureMoments":
        f = features.detach().flatten(start_dim=1).double()
        if self.limit is not None:
            f = f[:max(0, self.limit - self.n)]
        if f.shape[0] == 0:
            return self
        if self.s1 is None:
            self.s1 = torch.zeros(f.shape[1], dtype=torch.float64, device=f.device)
            self.s2 = torch.zeros(f.shape[1], f.shape[1], dtype=torch.float64, device=f.device)
        self.s1 += f.sum(dim=0)
        self.s2.addmm_(f.t(), f)
        self.n += f.shape[0]
        return self

    def all_reduce_(self) -> "FeatureMoments":
        """Sum (n, s1, s2) over the ranks of the default process group: the moments are additive, so every rank ends with
        the statistics of all rows any rank folded in (one [d, d] float64 all-reduce)."""
        if _ranks() > 1:
            if self.s1 is None:
                raise ValueError("a rank without feature rows cannot join the exchange (its feature width is unknown)")
            n = torch.tensor([float(self.n)], dtype=torch.float64, device=self.s1.device)
            for t in (n, self.s1, self.s2):
                torch_dist.all_reduce(t)
            self.n = int(n.item())
        return self

    def mean_cov(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Mean and the N - 1 normalised covariance (``np.cov(..., rowvar=False)``, validation_metrics.py:201-205)."""
        if self.n < 2:
            raise ValueError("a covariance needs at least two feature rows")
        mu = self.s1 / self.n
        cov = (self.s2 - self.n * torch.outer(mu, mu)) / (self.n - 1)
        return mu, cov


def _sqrt_product_trace(cov_a: torch.Tensor, cov_b: torch.Tensor) -> torch.Tensor:
    """tr(sqrtm(A B)) for symmetric positive semi-definite A, B."""
    cov_a, cov_b = 0.5 * (cov_a + cov_a.t()), 0.5 * (cov_b + cov_b.t())
    lam, q = torch.linalg.eigh(cov_a)
    root_a = (q * lam.clamp_min(0.0).sqrt()) @ q.t()
    inner = root_a @ cov_b @ root_a
    return torch.linalg.eigvalsh(0.5 * (inner + inner.t())).clamp_min(0.0).sqrt().sum()


def frechet_distance_from_moments(real: FeatureMoments, fake: FeatureMoments) -> float:
    (mu_r, cov_r), (mu_f, cov_f) = real.mean_cov(), fake.mean_cov()
    assert mu_r.shape == mu_f.shape and cov_r.shape == cov_f.shape
    diff = mu_r - mu_f
    value = diff @ diff + torch.trace(cov_r) + torch.trace(cov_f) - 2.0 * _sqrt_product_trace(cov_r, cov_f)
    return float(value)


def frechet_distance(real_activations: torch.Tensor, fake_activations: torch.Tensor) -> float:
    """FID._calc_fid / FVD._calc_fvd (validation_metrics.py:192-220, 401-429) on ``[samples, features]`` tensors, computed on
    the tensors' device."""
    return frechet_distance_from_moments(FeatureMoments().update(torch.as_tensor(real_activations)),
                                         FeatureMoments().update(torch.as_tensor(fake_activations)))


def inception_score(predictions: torch.Tensor) -> float:
    """validation_metrics.py:126-140: ``exp(mean_i KL(p_i || mean_j p_j))`` of softmax outputs ``[samples, classes]``."""
    p = torch.as_tensor(predictions).double()
    p_y = p.mean(dim=0, keepdim=True)
    kl = torch.sum(p * torch.log(p / p_y), dim=-1)
    return float(kl.mean().exp())


class _Metric:
    """Constructor of the three reference classes (validation_metrics.py:20-48, 162-189, 366-393) with the feature network
    as an argument.  ``data_parallel`` is accepted for signature compatibility and ignored: one process drives one GPU here
    (DESIGN.md section 6)."""

    def __init__(self, network: nn.Module, device: Union[str, torch.device] = "cuda", data_parallel: bool = False,
                 batch_size: int = 1, data_samples: int = 5000, no_rfp: bool = False, no_gfp: bool = False) -> None:
        if network is None:
            raise ValueError(f"{type(self).__name__} needs its feature network: the pretrained weights the reference loads "
                             "(torchvision Inception-v3 / I3D) are not part of this repository")
        self.network = network
        self.device, self.batch_size, self.data_samples = device, batch_size, data_samples
        self.no_rfp, self.no_gfp = no_rfp, no_gfp

    def _channels(self):
        return [0] + ([] if self.no_gfp else [1]) + ([] if self.no_rfp else [2])

    def _latents(self, generator):
        return misc.get_noise(batch_size=self.batch_size, latent_dimension=generator.latent_dimensions, p_mixed_noise=0.0,
                              device=self.device)

    def _result(self, scores):
        """The reference's return statements, in their order (validation_metrics.py:149-153, 352-358, 559-565): with a GFP
        channel the pair (bf, gfp) is returned whether or not RFP scores were computed."""
        if not self.no_gfp:
            return scores[0], scores[1]
        if not self.no_rfp:
            return tuple(scores)
        return scores[0]


class IS(_Metric):
    """Inception score of generated frames (validation_metrics.py:15-154).  ``network``: ``[B, 3, S, S]`` in [-1, 1] -> class
    logits (the reference: torchvision's Inception-v3).  Preprocessing as :44-52 -- bilinear, anti-aliased resize to
    ``input_size`` (299 x 299; the reference calls kornia 0.4.1's ``resize``, absent here: torch's own anti-aliased
    interpolation stands in, PARITY UNPINNED for that one call), then the batch-wise [-1, 1] normalisation."""

    def __init__(self, *args, input_size: Optional[Tuple[int, int]] = (299, 299), **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.input_size = input_size

    def _preprocessing(self, frames: torch.Tensor) -> torch.Tensor:
        x = frames[:, :, 0]
        if self.input_size is not None and tuple(x.shape[-2:]) != tuple(self.input_size):
            x = nn.functional.interpolate(x.float(), size=self.input_size, mode="bilinear", align_corners=False,
                                          antialias=True)
        return misc.normalize_m1_1_batch(x[:, :, None])[:, :, 0]

    @torch.no_grad()
    def __call__(self, generator: nn.Module, **kwargs):
        net = self.network.to(self.device).eval()
        generator.to(self.device).eval()
        channels = self._channels()
        predictions = [[] for _ in channels]
        world = _ranks()
        share = math.ceil(self.data_samples / world)             # this rank's samples; the class probabilities are gathered
        for _ in range(math.ceil(share / self.batch_size)):
            fake_images = generator(input=self._latents(generator))
            for k, c in enumerate(channels):
                frames = self._preprocessing(select_frames(fake_images, c))
                predictions[k].append(net(frames).float().softmax(dim=1))
        rows = []
        for p in predictions:
            p = torch.cat(p)[:share].contiguous()
            if world > 1:
                parts = [torch.empty_like(p) for _ in range(world)]
                torch_dist.all_gather(parts, p)
                p = torch.cat(parts)
            rows.append(p[:self.data_samples])
        return self._result([inception_score(p) for p in rows])


class _Frechet(_Metric):
    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.moments_real = None                        # cached across calls like the reference's activations_real_*

    def _inputs(self, images: torch.Tensor, channel: int) -> torch.Tensor:
        raise NotImplementedError

    @torch.no_grad()
    def __call__(self, generator: nn.Module, dataset: Iterable):
        net = self.network.to(self.device).eval()
        channels = self._channels()
        share = math.ceil(self.data_samples / _ranks())          # rows this rank contributes (all of them in one process)
        if self.moments_real is None:
            moments = [FeatureMoments(share) for _ in channels]
            for real_images in dataset:
                real_images = real_images.to(self.device)
                for m, c in zip(moments, channels):
                    m.update(net(self._inputs(real_images, c)))
                if moments[0].full:
                    break
            moments = [m.all_reduce_() for m in moments]
            if moments[0].n < self.data_samples:
                # (the reference indexes its activation list up to data_samples and would use however many rows there are
                #  too, silently; the score is then NOT the nominal-sample-count metric)
                warnings.warn(f"{type(self).__name__}: the dataset held {moments[0].n} samples, fewer than data_samples = "
                              f"{self.data_samples}; the real statistics (cached from now on) are those of {moments[0].n} rows")
            self.moments_real = moments
        generator.to(self.device).eval()
        fake = [FeatureMoments(share) for _ in channels]
        for _ in range(math.ceil(share / self.batch_size)):
            fake_images = generator(input=self._latents(generator))
            for m, c in zip(fake, channels):
                m.update(net(self._inputs(fake_images, c)))
        fake = [m.all_reduce_() for m in fake]
        return self._result([frechet_distance_from_moments(r, f) for r, f in zip(self.moments_real, fake)])


class FID(_Frechet):
    """Frechet inception distance between dataset frames and generated frames (validation_metrics.py:157-358).
    ``network``: ``[B, 3, H, W]`` in [-1, 1] -> pooled features (the reference: Inception-v3 up to the last pooling, 2048)."""

    def _inputs(self, images, channel):
        return misc.normalize_m1_1_batch(select_frames(images, channel))[:, :, 0]


class FVD(_Frechet):
    """Frechet video distance between dataset sequences and generated sequences (validation_metrics.py:361-568).
    ``network``: ``[B, 3, T, H, W]`` in [-1, 1] -> features of any shape (flattened per sample, :466; the reference: I3D)."""

    def _inputs(self, images, channel):
        return misc.normalize_m1_1_batch(select_clips(images, channel))
