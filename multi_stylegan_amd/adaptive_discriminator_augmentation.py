"""Adaptive discriminator augmentation on the MI355X, with the reference's class names and constructor / forward
signatures (multi_stylegan/adaptive_discriminator_augmentation.py:11-213).

What differs from the reference is mechanics only:

* the seven augmentation stages run as whole-batch device operations -- flip and integer translation as masked selects,
  the five kornia warps (90-degree rotation, isotropic scaling, rotation, anisotropic scaling, rotation) as ONE launch
  each of ``msg_affine_warp`` -- instead of Python loops over ``torch.rand`` results, index lists and
  ``images[idx] = ...`` assignments (:116-199), which cost a device->host synchronisation per stage;
* the augmentation probability ``p`` and the overfitting statistic ``r`` live on the device: the per-image selections
  are evaluated inside the kernels from the device-resident ``p``, ``r`` is accumulated without ``.item()`` (:47-48),
  and the +-p_step update every ``r_update`` fake batches is three tensor operations.  Under data parallelism ``r`` is
  averaged over the ranks (the reference computes it on the batch gathered on device 0);
* ``p`` and the pending ``r`` survive a checkpoint (``ada_state`` / ``load_ada_state``; the reference loses them).

**Parity unpinned**: kornia 0.4.1, whose arithmetic the warps restate, is not available (see oracle/ada.py).
"""
import math
import random
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from . import dist as msg_dist

SIGMA = (0.2 * math.log(2)) ** 2        # sigma the reference passes to np.random.lognormal (:142, :177)


def _warp_launch(x, angle, angle_const, scale, u, p, rot_prob, center, padding, align_corners, backward):
    dev = _lib.require_gpu(x, angle, scale, u, p)
    b, c, h, w = x.shape
    y = torch.empty_like(x)
    # backward: the scatter accumulates in 64-bit fixed point (deterministic whatever the order of the atomics)
    # (+ one word that a non-finite / out-of-range contribution raises: the gradient then comes out NaN, not wrapped)
    ws = torch.empty(x.numel() + 1, dtype=torch.int64, device=dev) if backward else None
    with _lib.on_device(dev), _lib.kernel_clock.span("affine_warp/f32", 2 * x.numel() * 4):
        code = _lib.lib().msg_affine_warp(x.data_ptr(), y.data_ptr(), _lib.ptr(angle), float(angle_const), _lib.ptr(scale),
                                          u.data_ptr(), p.data_ptr(), int(rot_prob), float(center[0]), float(center[1]),
                                          int(padding), int(align_corners), b, c, h, w, int(backward), _lib.ptr(ws),
                                          _lib.stream_of(dev))
    _lib.check(code, "msg_affine_warp")
    return y


class _AffineWarp(Function):
    """y = warp(x); differentiable in x (the generator's images pass through the augmentations, :66-70)."""

    @staticmethod
    def forward(ctx, x, angle, angle_const, scale, u, p, rot_prob, center, padding, align_corners):
        ctx.args = (angle, angle_const, scale, u, p, rot_prob, center, padding, align_corners)
        return _warp_launch(x, *ctx.args, backward=False)

    @staticmethod
    def backward(ctx, gy):
        gx = _warp_launch(gy.contiguous(), *ctx.args, backward=True)
        return (gx,) + (None,) * 9


def affine_warp(x: torch.Tensor, u: torch.Tensor, p: torch.Tensor, *, angle: Optional[torch.Tensor] = None,
                angle_const: float = 0.0, scale: Optional[torch.Tensor] = None, rot_prob: bool = False,
                center: Tuple[float, float], padding: int, align_corners: bool) -> torch.Tensor:
    """One warp stage over a batch [B, C, H, W] fp32: image b is warped iff u[b] <= p (or 1 - sqrt(1 - p))."""
    if x.dtype != torch.float32:
        raise _lib.MsgHipError("affine_warp: images are fp32")
    f32 = lambda t: None if t is None else t.to(x.device, torch.float32).contiguous()
    return _AffineWarp.apply(x.contiguous(), f32(angle), angle_const, f32(scale), f32(u), f32(p).reshape(1), rot_prob,
                             center, padding, align_corners)


def draw_augmentation(n: int, height: int, width: int, device) -> Dict[str, object]:
    """The random inputs of one pipeline call, drawn for every image from the reference's RNG sources in its order:
    torch.rand for the seven selections, random.choice for the 90-degree angle (:122), random.uniform for the integer
    translation (:209-211), numpy for the scales and angles (:141-143, :155-156, :176-178, :190-191).  Host-side numbers
    are uploaded with non-blocking copies; nothing here waits for the device."""
    host = {
        "u": torch.rand(7, n),
        "angle90": random.choice([0., -90., 90., 180.]),
        "roll": (int(height * random.uniform(-0.125, 0.125)), int(width * random.uniform(-0.125, 0.125))),
        "scale_iso": torch.from_numpy(np.random.lognormal(mean=0, sigma=SIGMA, size=(n, 1))).float().expand(n, 2),
        "angle_a": torch.from_numpy(np.random.uniform(low=-180, high=180, size=n)).float(),
        "scale_aniso": torch.from_numpy(np.random.lognormal(mean=0, sigma=SIGMA, size=(n, 2))).float(),
        "angle_b": torch.from_numpy(np.random.uniform(low=-180, high=180, size=n)).float(),
    }
    return {k: (v.contiguous().to(device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in host.items()}


class AugmentationPipeline(nn.Module):
    """The differentiable augmentation pipeline (reference :99-200)."""

    def forward(self, images: torch.Tensor, p: Union[float, torch.Tensor], draws: Optional[Dict[str, object]] = None
                ) -> torch.Tensor:
        """images [B, C, H, W] fp32 on the GPU; p: probability (a device scalar tensor keeps the call sync-free).
        ``draws``: see draw_augmentation (explicit for parity tests)."""
        n, _, h, w = images.shape
        dev = images.device
        if not isinstance(p, torch.Tensor):
            p = torch.tensor(float(p), dtype=torch.float32)
        p = p.to(dev, torch.float32)
        dr = draws if draws is not None else draw_augmentation(n, h, w, dev)
        u = dr["u"].to(dev)
        pick = lambda i: (u[i] <= p).view(n, 1, 1, 1)
        x = torch.where(pick(0), images.flip(dims=(-1,)), images)                                    # :116-118
        x = affine_warp(x, u[1], p, angle_const=dr["angle90"], center=((w - 1) / 2.0, (h - 1) / 2.0), padding=0,
                        align_corners=False)                                                          # :120-125 kaf.rotate
        x = torch.where(pick(2), torch.roll(x, shifts=dr["roll"], dims=(-2, -1)), x)                # :127-129
        mid = (0.5 * h, 0.5 * w)       # "center": 0.5 * images.shape[2:], assigned to (x, y) as it stands (:136-137)
        warp = lambda t, i, **kw: affine_warp(t, u[i], p, center=mid, padding=2, align_corners=True, **kw)
        x = warp(x, 3, scale=dr["scale_iso"])                                                         # :131-147
        x = warp(x, 4, angle=-dr["angle_a"], rot_prob=True)          # get_affine_matrix2d negates the angle  :149-165
        x = warp(x, 5, scale=dr["scale_aniso"])                                                       # :166-182
        x = warp(x, 6, angle=-dr["angle_b"], rot_prob=True)                                           # :184-199
        return x


def integer_translation(images: torch.Tensor) -> torch.Tensor:
    """reference :203-213: one random shift of up to an eighth of the size, applied with torch.roll."""
    shift = (int(images.shape[-2] * random.uniform(-0.125, 0.125)), int(images.shape[-1] * random.uniform(-0.125, 0.125)))
    return torch.roll(images, shifts=shift, dims=(-2, -1))


class AdaptiveDiscriminatorAugmentation(nn.Module):
    """Wraps a discriminator (reference :11-96): augments its input with probability ``p`` and adapts ``p`` so that the
    overfitting heuristic ``r = E[sign(D(fake))]`` stays at ``r_target``."""

    # forward(cat([real, fake]), minibatch_groups=2): the two halves are augmented with their own draws, go through the
    # discriminator as ONE batch with per-half minibatch statistics, and only the fake half feeds the controller --
    # the same result as the reference's two calls (real, then fake) at the launch count of one
    supports_minibatch_groups = True
    augments_in_place = True

    def __init__(self, discriminator: nn.Module, r_target: float = 0.6, p_step: float = 5e-03, r_update: int = 8,
                 p_max: float = 0.8) -> None:
        super().__init__()
        self.discriminator = discriminator
        self.r_target, self.p_step, self.r_update, self.p_max = r_target, p_step, r_update, p_max
        self.augmentation_pipeline = AugmentationPipeline()
        # controller state on the device, deliberately NOT registered buffers: the reference's state_dict holds the
        # discriminator only, and checkpoints stay interchangeable (ada_state / load_ada_state carry these)
        self._p = torch.tensor(0.05, dtype=torch.float32)
        self._r_sum = torch.zeros((), dtype=torch.float32)
        self._r_count = 0
        self.r_history: List[torch.Tensor] = []

    # the reference exposes plain attributes p and r (:37-38); reading p here costs a device->host copy
    @property
    def p(self) -> float:
        return float(self._p)

    @p.setter
    def p(self, value: float) -> None:
        self._p = torch.tensor(float(value), dtype=torch.float32, device=self._p.device)

    @property
    def compute_dtype(self):
        return self.discriminator.compute_dtype

    @compute_dtype.setter
    def compute_dtype(self, value):
        self.discriminator.compute_dtype = value

    def ada_state(self) -> Dict[str, object]:
        return {"p": float(self._p), "r_sum": float(self._r_sum), "r_count": self._r_count}

    def load_ada_state(self, state: Dict[str, object]) -> None:
        dev = self._p.device
        self._p = torch.tensor(float(state["p"]), dtype=torch.float32, device=dev)
        self._r_sum = torch.tensor(float(state.get("r_sum", 0.0)), dtype=torch.float32, device=dev)
        self._r_count = int(state.get("r_count", 0))

    def _observe(self, prediction_scalar: torch.Tensor, prediction_pixel_wise: torch.Tensor) -> None:
        """:46-48 and :76-94 without leaving the device."""
        with torch.no_grad():
            r = 0.5 * torch.sign(prediction_scalar).mean() + 0.5 * torch.sign(prediction_pixel_wise.mean(dim=(-1, -2))).mean()
            self._r_sum = self._r_sum.to(r.device) + r.float()
            self._r_count += 1
            if self._r_count >= self.r_update:
                r_mean = self._r_sum / self._r_count
                if msg_dist.collectives_active():
                    r_mean = msg_dist.all_reduce_mean(r_mean)        # the reference sees the batch gathered from all GPUs
                step = torch.where(r_mean > self.r_target, self.p_step, -self.p_step)
                self._p = (self._p.to(r.device) + step).clamp(0.0, self.p_max).float()
                self.r_history.append(r_mean)
                self._r_sum = torch.zeros_like(self._r_sum)
                self._r_count = 0

    def _augment(self, images: torch.Tensor, draws) -> torch.Tensor:
        shape = images.shape
        flat = images.flatten(start_dim=1, end_dim=2).float()
        augmented = self.augmentation_pipeline(flat, self._p, draws).view(shape)
        if not images.requires_grad and images.dtype == torch.float32:
            # the reference augments IN PLACE on a view of its argument (:64-68): whatever the caller does with the
            # batch afterwards in the same iteration (R1, CutMix) sees the augmented images
            with torch.no_grad():
                images.copy_(augmented)
        return augmented

    def forward(self, images: torch.Tensor, is_real: bool = False, is_cut_mix: bool = False,
                draws: Optional[Dict[str, object]] = None, minibatch_groups: int = 1, **kwargs
                ) -> Tuple[torch.Tensor, torch.Tensor]:
        if is_cut_mix:
            return self.discriminator(images, **kwargs)
        self._p = self._p.to(images.device)
        if minibatch_groups == 2:
            half = images.shape[0] // 2
            draws_real, draws_fake = draws if draws is not None else (None, None)
            augmented = torch.cat([self._augment(images[:half], draws_real), self._augment(images[half:], draws_fake)])
            prediction_scalar, prediction_pixel_wise = self.discriminator(augmented, minibatch_groups=2, **kwargs)
            self._observe(prediction_scalar[half:].detach(), prediction_pixel_wise[half:].detach())
            return prediction_scalar, prediction_pixel_wise
        if minibatch_groups != 1:
            raise ValueError("AdaptiveDiscriminatorAugmentation: one batch, or a (real, fake) pair with minibatch_groups=2")
        augmented = self._augment(images, draws)
        prediction_scalar, prediction_pixel_wise = self.discriminator(augmented, **kwargs)
        if not is_real:
            self._observe(prediction_scalar.detach(), prediction_pixel_wise.detach())
        return prediction_scalar, prediction_pixel_wise
