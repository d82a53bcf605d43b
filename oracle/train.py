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
}


def get_noise(batch_size, latent_dimension, p_mixed_noise=0.9, device="cpu"):
    """misc.py:238-252."""
    if p_mixed_noise > 0 and random.random() < p_mixed_noise:
        return list(torch.randn(2, batch_size, latent_dimension, device=device).unbind(0))
    return torch.randn(batch_size, latent_dimension, device=device)


def d_logistic_loss(pred_real, pred_fake):
    """loss.py:165-170."""
    return F.softplus(-pred_real).mean(), F.softplus(pred_fake).mean()


def g_logistic_loss(pred_fake):
    """loss.py:128-131."""
    return F.softplus(-pred_fake).mean()


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
                    trace: Optional[Dict] = None) -> Dict[str, float]:
    """One pass of model_wrapper.py:253-451; ``iteration`` is progress_bar.n (1-based, quirk Q12).  ``trace``: see
    ``clip_and_step``; labels d, r1, g, pl; the EMA movement goes to ``ema.delta.<param>``."""
    dr = draws or Draws()
    bsz, ld, dev = real.shape[0], g.latent_dimensions, real.device
    log: Dict[str, float] = {}
    # ---- D step (:260-305)
    _zero(opt_d, opt_g)
    with torch.no_grad():
        z = dr.z_d if dr.z_d is not None else get_noise(bsz, ld, hyper["p_mixed_noise"], dev)
        fake = g(z, inject_index=dr.inject_d, noise=dr.noise_d)
    pr, prp = d(real)
    pf, pfp = d(fake)
    l_r, l_f = d_logistic_loss(pr, pf)
    l_rp, l_fp = d_logistic_loss(prp, pfp)
    (l_r + l_f + l_rp + l_fp).backward()
    clip_and_step(d, opt_d, trace, "d")
    log.update(loss_d_real=l_r.item(), loss_d_fake=l_f.item(), loss_d_real_px=l_rp.item(),
               loss_d_fake_px=l_fp.item())
    # ---- lazy R1 (:307-329)
    if iteration % hyper["lazy_discriminator_regularization"] == 0:
        _zero(opt_d, opt_g)
        real_rg = real.detach().requires_grad_(True)
        pr, prp = d(real_rg)
        r1 = r1_penalty(pr, real_rg, prp)
        (hyper["w_discriminator_regularization_r1"] * r1).backward()
        clip_and_step(d, opt_d, trace, "r1")
        log["r1"] = r1.item()
    # ---- G step (:379-416)
    _zero(opt_d, opt_g)
    z = dr.z_g if dr.z_g is not None else get_noise(bsz, ld, hyper["p_mixed_noise"], dev)
    fake = g(z, inject_index=dr.inject_g, noise=dr.noise_g)
    pf, pfp = d(fake)
    l_g, l_gp = g_logistic_loss(pf), g_logistic_loss(pfp)
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
