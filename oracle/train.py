"""Oracle restatement of one adversarial training iteration (CPU, plain torch).

Follows multi_stylegan/model_wrapper.py:245-451 in the epoch-0 regime used by
the benchmark (CutMix / wrong-order / top-k inactive, ADA off): D step, lazy
R1, G step, lazy path-length regularisation, EMA.  All randomness can be
injected through ``draws`` so that the HIP trainer and this one see identical
inputs.
"""
import math
import random
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Union

import torch
import torch.nn.functional as F

HYPER = {  # multi_stylegan/config.py:30-57 (only the keys the hot path reads)
    "p_mixed_noise": 0.9,
    "lazy_generator_regularization": 16,
    "w_generator_regularization": math.log(2) / ((256 ** 2) * (math.log(256) - math.log(2))),
    "lazy_discriminator_regularization": 16,
    "w_discriminator_regularization_r1": 10.0,
    "batch_size_shrink_path_length_regularization": 0.5,
    "betas": (0.0, 0.999),
    "w_discriminator_regularization": 4.0,        # CutMix augmentation / consistency weight
    "batch_factor_wrong_order": 1. / 4.,
    "top_k_start": 1. / 4.,
    "top_k_finish": 3. / 4.,
    "wrong_order_start": 3. / 4.,
    "trap_weight": 1. / 4.,
}


def get_noise(batch_size, latent_dimension, p_mixed_noise=0.9, device="cpu"):
    """misc.py:238-252."""
    if p_mixed_noise > 0 and random.random() < p_mixed_noise:
        return list(torch.randn(2, batch_size, latent_dimension, device=device).unbind(0))
    return torch.randn(batch_size, latent_dimension, device=device)


def _weight_view(weight, like):
    return weight.view(1, 1, 1, weight.shape[-2], weight.shape[-1]).to(like.device)


def d_logistic_loss(pred_real, pred_fake, weight=None):
    """loss.py:146-170 (the optional trap-region weight map multiplies the per-pixel terms, :158-164)."""
    if weight is not None:
        return (F.softplus(-pred_real) * _weight_view(weight, pred_real)).mean(), \
            (F.softplus(pred_fake) * _weight_view(weight, pred_fake)).mean()
    return F.softplus(-pred_real).mean(), F.softplus(pred_fake).mean()


def g_logistic_loss(pred_fake, weight=None):
    """loss.py:109-131."""
    if weight is not None:
        return (F.softplus(-pred_fake) * _weight_view(weight, pred_fake)).mean()
    return F.softplus(-pred_fake).mean()


def d_logistic_loss_cut_mix(prediction, label):
    """loss.py:173-196: per-pixel real / fake terms selected by the binary CutMix label map."""
    return torch.mean(F.softplus(-prediction) * label), torch.mean(F.softplus(prediction) * (1. - label))


def random_permutation(n):
    """misc.py:202-213.  Despite the name the indices are drawn WITH replacement (np.random.choice(range(n),
    size=n)); only the identity is excluded (replaced by the reversal)."""
    import numpy as np
    permutation = torch.from_numpy(np.random.choice(range(n), size=n))
    if torch.equal(permutation, torch.arange(n)):
        permutation = torch.arange(start=n - 1, end=-1, step=-1)
    return permutation


def binary_cut_mix_map(height, width, device="cpu"):
    """u_net_2d_discriminator.py:425-448.  RNG consumption order (pinned by the golden test): torch.randint for the
    row cut, torch.randint for the column cut (both on the CPU generator), random.random() for the quadrant,
    random.random() for the inversion."""
    binary_map = torch.zeros(1, 1, 1, height, width, dtype=torch.float, device=device)
    cut_h = torch.randint(int(0.1 * height), int(0.9 * height), size=(1,))
    cut_w = torch.randint(int(0.1 * width), int(0.9 * width), size=(1,))
    if random.random() > 0.5:
        binary_map[..., cut_h:, cut_w:] = 1.0
    else:
        binary_map[..., :cut_h, :cut_w] = 1.0
    if random.random() > 0.5:
        binary_map = 1. - binary_map
    return binary_map


def cut_mix_augmentation_data(image_real, image_fake, binary_map=None):
    """u_net_2d_discriminator.py:384-399: real where the map is 1, fake elsewhere; the map is the label."""
    image_fake = image_fake[:image_real.shape[0]]
    if binary_map is None:
        binary_map = binary_cut_mix_map(image_real.shape[-2], image_fake.shape[-1], image_real.device)
    return image_real * binary_map + image_fake * (1. - binary_map), binary_map


def cut_mix_transformation_data(image_real, image_fake, prediction_real, prediction_fake, binary_map=None):
    """u_net_2d_discriminator.py:402-422: mixed image and the equally mixed pixel-wise predictions as soft target."""
    image_fake = image_fake[:image_real.shape[0]]
    prediction_fake = prediction_fake[:image_real.shape[0]]
    if binary_map is None:
        binary_map = binary_cut_mix_map(image_real.shape[-2], image_fake.shape[-1], image_real.device)
    return image_real * binary_map + image_fake * (1. - binary_map), \
        prediction_real * binary_map + prediction_fake * (1. - binary_map)


class TopK:
    """loss.py:398-444: keep the k = max(1, int(B v)) largest scalar predictions; v anneals linearly from 1 to 0.5
    between the two iteration marks.  Returns torch.topk's (values, indices)."""

    def __init__(self, starting_iteration, final_iteration):
        self.starting_iteration, self.final_iteration, self.iterations = starting_iteration, final_iteration, 0

    def calc_v(self):
        self.iterations += 1
        if self.iterations <= self.starting_iteration:
            return 1.
        if self.iterations >= self.final_iteration:
            return 0.5
        return 0.5 * (1. - float(self.iterations - self.starting_iteration)
                      / float(self.final_iteration - self.starting_iteration)) + 0.5

    def __call__(self, input):
        v = self.calc_v()
        input = input.view(-1)
        return torch.topk(input, k=max(1, int(input.shape[0] * v)))


def r1_penalty(pred_real, image_real, pred_real_pixel):
    """loss.py:311-316."""
    grad, = torch.autograd.grad((pred_real.sum(), pred_real_pixel.sum()), image_real, create_graph=True)
    return 0.5 * grad.pow(2).reshape(grad.shape[0], -1).sum(1).mean()


class PathLength:
    """loss.py:353-395 (mean_path_length is a plain attribute, quirk Q10)."""

    def __init__(self, decay=0.01):
        self.decay = decay
        self.mean_path_length = torch.zeros(1)

    def __call__(self, grad):
        self.mean_path_length = self.mean_path_length.detach().to(grad.device)
        lengths = torch.sqrt(grad.pow(2).sum(2).mean(1) + 1e-8).mean()
        self.mean_path_length = self.mean_path_length + self.decay * (lengths.mean() - self.mean_path_length)
        return torch.mean((lengths - self.mean_path_length) ** 2), lengths


@torch.no_grad()
def ema_update(g_ema, g, decay=0.999):
    """misc.py:183-199 (parameters only, buffers untouched)."""
    src = dict(g.named_parameters())
    for name, p in g_ema.named_parameters():
        p.mul_(decay).add_(src[name].detach(), alpha=1 - decay)


def make_optimizers(g, d, lr_g=2e-4, lr_d=6e-4):
    """train_multi_stylegan.py:53-57: Adam(beta=(0,0.999)); mapping net at lr/100."""
    og = torch.optim.Adam(g.get_parameters(lr_main=lr_g, lr_style=lr_g / 100.), betas=HYPER["betas"])
    od = torch.optim.Adam(d.parameters(), lr=lr_d, betas=HYPER["betas"])
    return og, od


@dataclass
class Draws:
    """Explicit random inputs for one iteration (None = draw inside)."""
    z_d: Optional[Union[torch.Tensor, List[torch.Tensor]]] = None
    z_g: Optional[Union[torch.Tensor, List[torch.Tensor]]] = None
    z_pl: Optional[Union[torch.Tensor, List[torch.Tensor]]] = None
    inject_d: Optional[int] = None
    inject_g: Optional[int] = None
    inject_pl: Optional[int] = None
    noise_d: Optional[List[torch.Tensor]] = None      # per-layer noise lists (len 1 + 2*levels)
    noise_g: Optional[List[torch.Tensor]] = None
    noise_pl: Optional[List[torch.Tensor]] = None
    pl_image_noise: Optional[torch.Tensor] = None     # replaces the randn of generator.py:195
    wrong_order_perm: Optional[torch.Tensor] = None   # replaces misc.random_permutation (model_wrapper.py:276)
    cut_mix: Optional[bool] = None                    # replaces the random.random() gate (model_wrapper.py:331-332)
    cut_mix_map_aug: Optional[torch.Tensor] = None    # replace the two random binary maps (:337, :357)
    cut_mix_map_reg: Optional[torch.Tensor] = None


def _pl_grads(g, z, inject, noise, image_noise):
    """Generator.forward(return_path_length_grads=True) with the image noise injectable."""
    if image_noise is None:
        return g(z, inject_index=inject, noise=noise, return_path_length_grads=True)
    image, latent = g(z, inject_index=inject, noise=noise, return_main_style_vectors=True)
    pn = image_noise / math.sqrt(image.shape[2] * image.shape[3] * image.shape[4])
    return torch.autograd.grad((image * pn).sum(), latent, create_graph=True, retain_graph=True)[0]


def _zero(*optimizers) -> None:
    """optimizer.zero_grad() as the reference's pinned torch 1.8.1 (requirements.txt:1) executes it: gradients are
    ZEROED, not dropped (set_to_none became the default only in torch 2.0).  It matters in the two regulariser steps:
    parameters the double-backward graph does not reach (e.g. the additive output-block biases in the path-length
    step) keep a zero gradient, so Adam still takes a (zero-length) step for them -- its step count advances and its
    second-moment estimate decays -- instead of skipping them."""
    for opt in optimizers:
        opt.zero_grad(set_to_none=False)


def clip_and_step(model, optimizer, trace: Optional[Dict], label: str) -> None:
    """clip_grad_norm_(5.) + optimizer.step() (model_wrapper.py:296-298, 325-326, 410-412, 440-441).  With a
    ``trace`` dict the step is recorded for parity tests: pre-clip gradients ``<label>.grad.<param>``, their global
    norm ``<label>.gnorm`` and the parameter movement ``<label>.delta.<param>``."""
    named = [(n, p) for n, p in model.named_parameters() if p.grad is not None]
    if trace is not None:
        before = {n: p.detach().clone() for n, p in named}
        for n, p in named:
            trace[f"{label}.grad.{n}"] = p.grad.detach().clone()
    total = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=5.)
    optimizer.step()
    if trace is not None:
        trace[f"{label}.gnorm"] = total.detach().clone()
        for n, p in named:
            trace[f"{label}.delta.{n}"] = p.detach() - before[n]


def train_iteration(g, d, g_ema, opt_g, opt_d, path_length: PathLength, real: torch.Tensor,
                    iteration: int, draws: Optional[Draws] = None, hyper: Dict = HYPER,
                    trace: Optional[Dict] = None, epoch: int = 0, epochs: int = 1, resume_training: bool = False,
                    top_k=None, trap_weights_map: Optional[torch.Tensor] = None) -> Dict[str, float]:
    """One pass of model_wrapper.py:253-451; ``iteration`` is progress_bar.n (1-based, quirk Q12).  ``trace``: see
    ``clip_and_step``; labels d, r1, cm_aug, cm_reg, g, pl; the EMA movement goes to ``ema.delta.<param>``.
    ``epoch`` / ``epochs`` / ``resume_training`` switch on the late-training branches exactly as the reference does:
    wrongly ordered reals among the fakes (:272-277), the trap-region weight map (:289-291, :404-406), CutMix
    augmentation + consistency regularisation (:331-376); ``top_k`` is the module of loss.py:398-444 (:392-401)."""
    dr = draws or Draws()
    bsz, ld, dev = real.shape[0], g.latent_dimensions, real.device
    log: Dict[str, float] = {}
    weight = trap_weights_map if (hyper["trap_weight"] * epochs <= epoch or resume_training) else None
    # ---- D step (:260-305)
    _zero(opt_d, opt_g)
    with torch.no_grad():
        z = dr.z_d if dr.z_d is not None else get_noise(bsz, ld, hyper["p_mixed_noise"], dev)
        fake = g(z, inject_index=dr.inject_d, noise=dr.noise_d)
    if epoch >= hyper["wrong_order_start"] * epochs or resume_training:                 # :272-277
        perm = dr.wrong_order_perm if dr.wrong_order_perm is not None else random_permutation(real.shape[2])
        fake = torch.cat([fake, real[:max(1, int(hyper["batch_factor_wrong_order"] * bsz)), :, perm]], dim=0)
    pr, prp = d(real)
    pf, pfp = d(fake)
    l_r, l_f = d_logistic_loss(pr, pf)
    l_rp, l_fp = d_logistic_loss(prp, pfp, weight)
    (l_r + l_f + l_rp + l_fp).backward()
    clip_and_step(d, opt_d, trace, "d")
    log.update(loss_d_real=l_r.item(), loss_d_fake=l_f.item(), loss_d_real_px=l_rp.item(),
               loss_d_fake_px=l_fp.item())
    # ---- lazy R1 (:307-329)
    real_in = real
    if iteration % hyper["lazy_discriminator_regularization"] == 0:
        _zero(opt_d, opt_g)
        real_in = real.detach().requires_grad_(True)          # :313 -- and it STAYS that way for the CutMix block
        pr, prp = d(real_in)                                  # overwrites the D step's real predictions (:315)
        r1 = r1_penalty(pr, real_in, prp)
        (hyper["w_discriminator_regularization_r1"] * r1).backward()
        clip_and_step(d, opt_d, trace, "r1")
        log["r1"] = r1.item()
    # ---- CutMix augmentation + consistency regularisation (:331-376)
    if dr.cut_mix is not None:
        do_cut_mix = dr.cut_mix
    else:
        do_cut_mix = (random.random() <= (0.5 / float(epochs)) * float(epoch)) or \
            (resume_training and random.random() <= 0.5)
    if do_cut_mix:
        _zero(opt_d, opt_g)
        cm_images, cm_label = cut_mix_augmentation_data(real_in, fake, dr.cut_mix_map_aug)
        _, cm_pred = d(cm_images)
        cm_real, cm_fake = d_logistic_loss_cut_mix(cm_pred, cm_label)
        (hyper["w_discriminator_regularization"] * (cm_real + cm_fake)).backward()
        clip_and_step(d, opt_d, trace, "cm_aug")
        log["cut_mix_aug"] = (cm_real + cm_fake).item()
        _zero(opt_d)                                           # :355 -- the discriminator's optimizer only
        cr_images, cr_label = cut_mix_transformation_data(real_in.detach(), fake.detach(), prp.detach(),
                                                          pfp.detach(), dr.cut_mix_map_reg)
        _, cr_pred = d(cr_images)
        cr_loss = F.mse_loss(cr_pred, cr_label, reduction="mean")
        (hyper["w_discriminator_regularization"] * cr_loss).backward()
        clip_and_step(d, opt_d, trace, "cm_reg")
        log["cut_mix_reg"] = cr_loss.item()
    # ---- G step (:379-416)
    _zero(opt_d, opt_g)
    z = dr.z_g if dr.z_g is not None else get_noise(bsz, ld, hyper["p_mixed_noise"], dev)
    fake = g(z, inject_index=dr.inject_g, noise=dr.noise_g)
    pf, pfp = d(fake)
    if top_k is not None:                                      # :392-401
        pf, indexes = top_k(pf)
        pfp = pfp[indexes]
    l_g, l_gp = g_logistic_loss(pf), g_logistic_loss(pfp, weight)
    (l_g + l_gp).backward()
    clip_and_step(g, opt_g, trace, "g")
    log.update(loss_g=l_g.item(), loss_g_px=l_gp.item())
    # ---- lazy path length (:418-444)
    if iteration % hyper["lazy_generator_regularization"] == 0:
        _zero(opt_d, opt_g)
        n_pl = max(1, int(hyper["batch_size_shrink_path_length_regularization"] * bsz))
        z = dr.z_pl if dr.z_pl is not None else get_noise(n_pl, ld, hyper["p_mixed_noise"], dev)
        grads = _pl_grads(g, z, dr.inject_pl, dr.noise_pl, dr.pl_image_noise)
        pl_loss, pl_len = path_length(grads)
        (hyper["w_generator_regularization"] * pl_loss).backward()
        clip_and_step(g, opt_g, trace, "pl")
        log.update(path_length=pl_len.mean().item(), loss_pl=pl_loss.item())
    if trace is not None:
        ema_before = {n: p.detach().clone() for n, p in g_ema.named_parameters()}
    ema_update(g_ema, g)                                            # :446
    if trace is not None:
        for n, p in g_ema.named_parameters():
            trace[f"ema.delta.{n}"] = p.detach() - ema_before[n]
    return log
