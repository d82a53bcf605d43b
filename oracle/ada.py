"""Oracle restatement of adaptive discriminator augmentation (CPU, plain torch) -- TEST INFRASTRUCTURE.

Follows multi_stylegan/adaptive_discriminator_augmentation.py:11-213.  **Parity unpinned**: the reference's
arithmetic for five of the seven stages lives in kornia 0.4.1 (``requirements.txt:8``; ``kaf.rotate`` at :124,
``kaf.apply_affine`` at :133, :152, :168, :187), which is neither vendored under /root/reference nor installed
here, and the reference has no tests or golden vectors for this path.  What is restated below is kornia 0.4.1's
published algorithm as the reference calls it:

* ``apply_affine(input, params, flags)``: ``get_affine_matrix2d(translations, center, scale, angle, sx, sy)`` =
  ``get_rotation_matrix2d(center, -angle, scale)`` (+ translations, no shear here), then
  ``warp_affine(input, M, (H, W), mode, padding_mode, align_corners)``: the 3x3 matrix is conjugated with
  ``normal_transform_pixel`` (pixel [0, size-1] -> [-1, 1]), inverted, turned into a sampling grid by
  ``F.affine_grid`` and applied with ``F.grid_sample`` -- flags resample=1 (bilinear), padding_mode=2 (reflection),
  align_corners=True at every call site;
* ``get_rotation_matrix2d(center, angle, scale)``: ``R(angle) @ diag(scale)`` with
  ``R = [[cos, sin], [-sin, cos]]`` (degrees), translation column ``((1 - a) cx - b cy, b cx + (1 - a) cy)`` with
  ``a, b`` the first row of the scaled rotation;
* ``rotate(tensor, angle)`` (the 90-degree stage): the same warp about the tensor centre ``((W-1)/2, (H-1)/2)`` with
  bilinear sampling, zeros padding and kornia 0.4.1's default ``align_corners=False``.

Every random quantity is an explicit input (``Draws``) so that the HIP pipeline and this one can be compared on
identical draws; ``draw()`` produces them in the reference's order from its RNG sources.
"""
import math
import random
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch
import torch.nn.functional as F

SIGMA = (0.2 * math.log(2)) ** 2          # the reference passes sigma = (0.2 ln 2)^2 to np.random.lognormal (:142, :177)


@dataclass
class Draws:
    """Random inputs of one AugmentationPipeline.forward call for a batch of n images.  The u_* are the uniform
    numbers compared with p (or 1 - sqrt(1 - p)) to select the images a stage touches."""
    u_flip: torch.Tensor            # [n]
    u_rot90: torch.Tensor           # [n]
    angle90: float                  # one of 0, -90, 90, 180 -- ONE angle for all selected images (:122)
    u_roll: torch.Tensor            # [n]
    roll: tuple                     # (rows, cols) integer shift, one for all selected images (:209-211)
    u_iso: torch.Tensor             # [n]
    scale_iso: torch.Tensor         # [n]      lognormal
    u_rot_a: torch.Tensor           # [n]
    angle_a: torch.Tensor           # [n]      uniform(-180, 180)
    u_aniso: torch.Tensor           # [n]
    scale_aniso: torch.Tensor       # [n, 2]   lognormal
    u_rot_b: torch.Tensor           # [n]
    angle_b: torch.Tensor           # [n]


def draw(n: int, height: int, width: int) -> Draws:
    """All draws of one call, every stage drawn for every image (the reference draws parameters only for the
    selected ones; the selection itself is what `u_* <= p` decides)."""
    return Draws(
        u_flip=torch.rand(n), u_rot90=torch.rand(n), angle90=random.choice([0., -90., 90., 180.]),
        u_roll=torch.rand(n), roll=(int(height * random.uniform(-0.125, 0.125)), int(width * random.uniform(-0.125, 0.125))),
        u_iso=torch.rand(n), scale_iso=torch.from_numpy(np.random.lognormal(mean=0, sigma=SIGMA, size=(n,))).float(),
        u_rot_a=torch.rand(n), angle_a=torch.from_numpy(np.random.uniform(low=-180, high=180, size=n)).float(),
        u_aniso=torch.rand(n), scale_aniso=torch.from_numpy(np.random.lognormal(mean=0, sigma=SIGMA, size=(n, 2))).float(),
        u_rot_b=torch.rand(n), angle_b=torch.from_numpy(np.random.uniform(low=-180, high=180, size=n)).float())


def rotation_matrix2d(center: torch.Tensor, angle_deg: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """kornia 0.4.1 get_rotation_matrix2d: center [n,2] (x, y), angle [n] degrees, scale [n,2] -> [n,2,3]."""
    rad = angle_deg * math.pi / 180.0
    cos, sin = torch.cos(rad), torch.sin(rad)
    rot = torch.stack([cos, sin, -sin, cos], dim=-1).view(-1, 2, 2)
    scaled = rot @ torch.diag_embed(scale)
    alpha, beta = scaled[:, 0, 0], scaled[:, 0, 1]
    x, y = center[:, 0], center[:, 1]
    m = torch.zeros(angle_deg.shape[0], 2, 3, dtype=center.dtype)
    m[:, :, :2] = scaled
    m[:, 0, 2] = (1.0 - alpha) * x - beta * y
    m[:, 1, 2] = beta * x + (1.0 - alpha) * y
    return m


def _normal_transform_pixel(height: int, width: int) -> torch.Tensor:
    """kornia normal_transform_pixel: pixel coordinates [0, size-1] -> [-1, 1]."""
    return torch.tensor([[2.0 / (width - 1), 0.0, -1.0], [0.0, 2.0 / (height - 1), -1.0], [0.0, 0.0, 1.0]])


def warp_affine(x: torch.Tensor, m: torch.Tensor, padding_mode: str, align_corners: bool) -> torch.Tensor:
    """kornia 0.4.1 warp_affine(src, M, dsize=(H, W), 'bilinear', padding_mode, align_corners)."""
    n, _, height, width = x.shape
    m3 = torch.cat([m, torch.tensor([[[0.0, 0.0, 1.0]]]).expand(n, 1, 3)], dim=1)
    norm = _normal_transform_pixel(height, width)
    dst_norm_trans_src_norm = norm @ m3 @ torch.inverse(norm)
    theta = torch.inverse(dst_norm_trans_src_norm)[:, :2, :]
    grid = F.affine_grid(theta, [n, x.shape[1], height, width], align_corners=align_corners)
    return F.grid_sample(x, grid, mode="bilinear", padding_mode=padding_mode, align_corners=align_corners)


def _affine_stage(images, select, angle_deg, scale):
    """images[select] = kaf.apply_affine(images[select], ...): centre 0.5 * (H, W) assigned as (x, y) (:136-137),
    get_affine_matrix2d negates the angle, bilinear / reflection / align_corners=True (:145-147)."""
    idx = torch.nonzero(select).flatten()
    if idx.numel() == 0:
        return images
    h, w = images.shape[2:]
    center = torch.ones(idx.numel(), 2) * 0.5 * torch.tensor([float(h), float(w)])
    m = rotation_matrix2d(center, -angle_deg[idx], scale[idx])
    out = images.clone()
    out[idx] = warp_affine(images[idx], m, "reflection", True)
    return out


def augment(images: torch.Tensor, p: float, dr: Draws) -> torch.Tensor:
    """AugmentationPipeline.forward (:107-200) on images [n, C, H, W]; returns the augmented batch (the reference
    works in place on its argument)."""
    n, _, h, w = images.shape
    p_rot = 1.0 - math.sqrt(1.0 - p)
    x = images
    sel = dr.u_flip <= p                                               # :116-118  flip along the width
    x = torch.where(sel.view(n, 1, 1, 1), x.flip(dims=(-1,)), x)
    sel = dr.u_rot90 <= p                                              # :120-125  rotation by one multiple of 90 degrees
    idx = torch.nonzero(sel).flatten()
    if idx.numel() > 0:
        center = torch.tensor([[(w - 1) / 2.0, (h - 1) / 2.0]]).expand(idx.numel(), 2)
        m = rotation_matrix2d(center, torch.full((idx.numel(),), float(dr.angle90)), torch.ones(idx.numel(), 2))
        x = x.clone()
        x[idx] = warp_affine(x[idx], m, "zeros", False)
    sel = dr.u_roll <= p                                               # :127-129  integer translation (torch.roll)
    x = torch.where(sel.view(n, 1, 1, 1), torch.roll(x, shifts=dr.roll, dims=(-2, -1)), x)
    zeros = torch.zeros(n)
    x = _affine_stage(x, dr.u_iso <= p, zeros, dr.scale_iso.view(n, 1).expand(n, 2))          # :131-147 isotropic scaling
    x = _affine_stage(x, dr.u_rot_a <= p_rot, dr.angle_a, torch.ones(n, 2))                   # :149-165 rotation
    x = _affine_stage(x, dr.u_aniso <= p, zeros, dr.scale_aniso)                              # :166-182 anisotropic scaling
    x = _affine_stage(x, dr.u_rot_b <= p_rot, dr.angle_b, torch.ones(n, 2))                   # :184-199 rotation
    return x


class Controller:
    """The p controller of AdaptiveDiscriminatorAugmentation (:37-39, :46-48, :76-94): r = 0.5 mean sign(D scalar) +
    0.5 mean sign(mean pixel-wise) per FAKE batch; every r_update fake batches p moves by +-p_step towards
    r_target and is clamped to [0, p_max]."""

    def __init__(self, r_target=0.6, p_step=5e-3, r_update=8, p_max=0.8):
        self.r_target, self.p_step, self.r_update, self.p_max = r_target, p_step, r_update, p_max
        self.r: List[float] = []
        self.p = 0.05
        self.r_history: List[float] = []

    def observe(self, prediction_scalar: torch.Tensor, prediction_pixel_wise: torch.Tensor, is_real: bool) -> None:
        if not is_real:
            self.r.append((0.5 * torch.mean(torch.sign(prediction_scalar))
                           + 0.5 * torch.mean(torch.sign(prediction_pixel_wise.mean(dim=(-1, -2))))).item())
        if len(self.r) >= self.r_update:
            r = float(np.mean(self.r))
            self.p += self.p_step if r > self.r_target else -self.p_step
            self.p = self.p if self.p >= 0. else 0.
            self.p = self.p if self.p < self.p_max else self.p_max
            self.r = []
            self.r_history.append(r)
