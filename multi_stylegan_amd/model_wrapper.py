"""The adversarial training step on MI355X: the hot loop of the reference's ``ModelWrapper._gan_training``
(multi_stylegan/model_wrapper.py:245-451) as a data-parallel, sync-free iteration.

Per iteration, exactly the reference's sequence in its epoch-0 regime (CutMix, wrong-order augmentation and top-k
are inactive there, ADA is off): D step on a no-grad fake batch -> lazy R1 every 16th iteration -> G step through
D -> lazy path-length regularisation on half a batch every 16th iteration -> EMA.  Each optimiser step is
clip(5.0) + Adam(beta=(0, 0.999)).

What differs from the reference is only mechanics: one process per GPU with bucketed RCCL gradient averaging
overlapped with backward (``dist.GradBucketReducer``), gradient zeroing / clipping on flat buffers, losses kept on
the device (the reference calls ``.item()`` ten times per iteration), fused multi-tensor Adam and EMA.
"""
import copy
import math
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Union

import torch
import torch.nn as nn

from . import dist as msg_dist
from . import loss, misc
from .config import generation_hyperparameters


@dataclass
class Draws:
    """Explicit random inputs of one iteration (for parity runs); any field left None is drawn on the fly."""
    z_d: Optional[Union[torch.Tensor, List[torch.Tensor]]] = None
    z_g: Optional[Union[torch.Tensor, List[torch.Tensor]]] = None
    z_pl: Optional[Union[torch.Tensor, List[torch.Tensor]]] = None
    inject_d: Optional[int] = None
    inject_g: Optional[int] = None
    inject_pl: Optional[int] = None
    noise_d: Optional[List[torch.Tensor]] = None
    noise_g: Optional[List[torch.Tensor]] = None
    noise_pl: Optional[List[torch.Tensor]] = None
    pl_image_noise: Optional[torch.Tensor] = None

    def to(self, device):
        def mv(v):
            if v is None or isinstance(v, int):
                return v
            if isinstance(v, (list, tuple)):
                return [mv(t) for t in v]
            return v.to(device)
        return Draws(**{k: mv(v) for k, v in self.__dict__.items()})


class ModelWrapper(object):
    """Owns G, D, the EMA copy, both optimisers and the regulariser state, and runs training iterations.

    Constructor arguments follow the reference's wrapper where they concern the hot path; dataset, logger and
    validation metrics are outside it (the caller feeds ``train_iteration`` with batches).
    """

    def __init__(self, generator: nn.Module, discriminator: nn.Module,
                 generator_optimizer: Optional[torch.optim.Optimizer] = None,
                 discriminator_optimizer: Optional[torch.optim.Optimizer] = None,
                 hyperparameters: Dict[str, Any] = generation_hyperparameters,
                 generator_loss: nn.Module = None, discriminator_loss: nn.Module = None,
                 discriminator_regularization_loss: nn.Module = None,
                 path_length_regularization: nn.Module = None, generator_ema: Optional[nn.Module] = None,
                 device: str = "cuda", lr_generator: float = 2e-4, lr_discriminator: float = 6e-4,
                 bucket_bytes: int = 32 << 20, overlap_communication: bool = True,
                 skip_discriminator_weight_grads_in_generator_step: bool = False,
                 fused_optimizer: Optional[bool] = None, batch_discriminator_passes: bool = True) -> None:
        self.device = torch.device(device)
        self.generator = generator.to(self.device)
        self.discriminator = discriminator.to(self.device)
        self.hyperparameters = hyperparameters
        self.generator_loss = generator_loss or loss.NonSaturatingLogisticGeneratorLoss()
        self.discriminator_loss = discriminator_loss or loss.NonSaturatingLogisticDiscriminatorLoss()
        self.discriminator_regularization_loss = discriminator_regularization_loss or loss.R1Regularization()
        self.path_length_regularization = (path_length_regularization or loss.PathLengthRegularization()) \
            .to(self.device)
        # identical replicas on every rank, then the EMA copy (reference :81-90)
        msg_dist.broadcast_module(self.generator)
        msg_dist.broadcast_module(self.discriminator)
        self.generator_ema = generator_ema if generator_ema is not None else copy.deepcopy(self.generator)
        self.generator_ema.to(self.device).eval().requires_grad_(False)
        self.latent_dimensions = self.generator.latent_dimensions
        if fused_optimizer is None:
            fused_optimizer = self.device.type == "cuda"
        betas = hyperparameters["betas"]
        # reference train_multi_stylegan.py:53-57
        self.generator_optimizer = generator_optimizer or torch.optim.Adam(
            self.generator.get_parameters(lr_main=lr_generator, lr_style=lr_generator / 100.), betas=betas,
            fused=fused_optimizer)
        self.discriminator_optimizer = discriminator_optimizer or torch.optim.Adam(
            self.discriminator.parameters(), lr=lr_discriminator, betas=betas, fused=fused_optimizer)
        live = self.generator.live_parameters() if hasattr(self.generator, "live_parameters") \
            else list(self.generator.parameters())
        self.generator_reducer = msg_dist.GradBucketReducer(live, bucket_bytes, overlap_communication)
        self.discriminator_reducer = msg_dist.GradBucketReducer(self.discriminator.parameters(), bucket_bytes,
                                                                overlap_communication)
        self.skip_d_wgrad = skip_discriminator_weight_grads_in_generator_step
        # only discriminators that know about minibatch groups (ours) can take the concatenated batch
        self.batch_discriminator_passes = batch_discriminator_passes and \
            getattr(discriminator, "supports_minibatch_groups", False)
        self.iteration = 0                       # == progress_bar.n of the reference (1-based when tested, Q12)
        self.step_trace: Optional[Dict[str, torch.Tensor]] = None      # see _step
        self._param_names = {id(p): n for mod in (self.generator, self.discriminator)
                             for n, p in mod.named_parameters()}
        self._log: Dict[str, List[torch.Tensor]] = {}

    # ------------------------------------------------------------------------------------------------ utilities
    def _record(self, **values: torch.Tensor) -> None:
        for key, value in values.items():
            self._log.setdefault(key, []).append(value.detach().float().reshape(()))

    def pop_logs(self) -> Dict[str, List[float]]:
        """One device->host transfer for everything recorded since the last call."""
        out = {}
        if self._log:
            keys = list(self._log)
            flat = torch.stack([v for k in keys for v in self._log[k]]).cpu().tolist()
            pos = 0
            for k in keys:
                n = len(self._log[k])
                out[k] = flat[pos:pos + n]
                pos += n
        self._log = {}
        return out

    def _noise(self, batch_size: int):
        return misc.get_noise(batch_size=batch_size, latent_dimension=self.latent_dimensions,
                              p_mixed_noise=self.hyperparameters["p_mixed_noise"], device=self.device)

    def _step(self, reducer: msg_dist.GradBucketReducer, optimizer: torch.optim.Optimizer, label: str = "") -> None:
        """finish the gradient exchange, clip to norm 5 (reference :296,:410), Adam step.  With torch's fused Adam the
        clip factor rides along as its `grad_scale` (the hook GradScaler uses: grad <- grad / grad_scale inside the
        optimizer kernel), which saves a read-modify-write pass over every gradient.

        ``self.step_trace`` (a dict, off by default) records the step for parity tests under ``<label>.``: the
        pre-clip mean gradient and the movement of every parameter the reducer owns, and the global gradient norm."""
        fused = isinstance(optimizer, torch.optim.Adam) and bool(optimizer.defaults.get("fused"))
        # fused: buckets keep rank SUMS; `pending` = 1 / world is still to be applied
        pending = reducer.finish(average=not fused)
        trace = self.step_trace
        if trace is not None:
            named = [(self._param_names[id(p)], p) for b in reducer.buckets for p in b.params]
            before = {n: p.detach().clone() for n, p in named}
            for n, p in named:
                trace[f"{label}.grad.{n}"] = p.grad.detach() * pending
        if fused:
            total = reducer.grad_norm() * pending          # norm of the mean gradient
            coef = torch.clamp(5.0 / (total + 1e-6), max=1.0) * pending
            optimizer.grad_scale = (1.0 / coef).reshape(()).float()
            optimizer.found_inf = torch.zeros((), dtype=torch.float32, device=total.device)
            try:
                optimizer.step()
            finally:
                del optimizer.grad_scale, optimizer.found_inf
        else:
            total = reducer.clip_(5.0)
            optimizer.step()
        if trace is not None:
            trace[f"{label}.gnorm"] = total.detach().clone()
            for n, p in named:
                trace[f"{label}.delta.{n}"] = p.detach() - before[n]

    def _zero(self) -> None:
        self.discriminator_reducer.zero_grad()
        self.generator_reducer.zero_grad()

    # ------------------------------------------------------------------------------------------- one iteration
    def train_iteration(self, real_images: torch.Tensor, draws: Optional[Draws] = None) -> None:
        hp = self.hyperparameters
        dr = draws or Draws()
        G, D = self.generator, self.discriminator
        self.iteration += 1
        real_images = real_images.to(self.device, non_blocking=True)
        batch = real_images.shape[0]
        # ---------------- discriminator step (reference :258-305)
        self._zero()
        with torch.no_grad():
            z = dr.z_d if dr.z_d is not None else self._noise(batch)
            fake_images = G(input=z, inject_index=dr.inject_d, noise=dr.noise_d)
        self.discriminator_reducer.arm()
        if self.batch_discriminator_passes and real_images.shape == fake_images.shape:
            # D(real) and D(fake) of the reference (:272-275) as ONE batch of 2B with per-half minibatch statistics:
            # same result, half the launches, better-filled tiles on the low-resolution layers
            both, both_px = D(torch.cat([real_images, fake_images.to(real_images.dtype)]), minibatch_groups=2)
            (real_prediction, fake_prediction) = both.split(batch)
            (real_prediction_pixel_wise, fake_prediction_pixel_wise) = both_px.split(batch)
        else:
            real_prediction, real_prediction_pixel_wise = D(real_images, is_real=True, is_cut_mix=False)
            fake_prediction, fake_prediction_pixel_wise = D(fake_images, is_real=False, is_cut_mix=False)
        loss_real, loss_fake = self.discriminator_loss(real_prediction, fake_prediction)
        loss_real_px, loss_fake_px = self.discriminator_loss(real_prediction_pixel_wise, fake_prediction_pixel_wise)
        (loss_real + loss_fake + loss_real_px + loss_fake_px).backward()
        self._step(self.discriminator_reducer, self.discriminator_optimizer, "d")
        self._record(loss_discriminator_real=loss_real, loss_discriminator_fake=loss_fake,
                     loss_discriminator_real_pixel_wise=loss_real_px,
                     loss_discriminator_fake_pixel_wise=loss_fake_px)
        # ---------------- lazy R1 (reference :307-329)
        if self.iteration % hp["lazy_discriminator_regularization"] == 0:
            self._zero()
            real_rg = real_images.detach().requires_grad_(True)
            self.discriminator_reducer.arm()
            real_prediction, real_prediction_pixel_wise = D(real_rg, is_real=False, is_cut_mix=True)
            r1 = self.discriminator_regularization_loss(real_prediction, real_rg, real_prediction_pixel_wise)
            (hp["w_discriminator_regularization_r1"] * r1).backward()
            self._step(self.discriminator_reducer, self.discriminator_optimizer, "r1")
            self._record(loss_discriminator_regularization=r1)
        # ---------------- generator step (reference :377-416)
        self._zero()
        z = dr.z_g if dr.z_g is not None else self._noise(batch)
        if self.skip_d_wgrad:
            D.requires_grad_(False)
        self.generator_reducer.arm()
        fake_images = G(input=z, inject_index=dr.inject_g, noise=dr.noise_g)
        fake_prediction, fake_prediction_pixel_wise = D(fake_images, is_real=False, is_cut_mix=False)
        loss_g = self.generator_loss(fake_prediction)
        loss_g_px = self.generator_loss(fake_prediction_pixel_wise)
        (loss_g + loss_g_px).backward()
        if self.skip_d_wgrad:
            D.requires_grad_(True)
        self._step(self.generator_reducer, self.generator_optimizer, "g")
        self._record(loss_generator=loss_g, loss_generator_pixel_wise=loss_g_px)
        # ---------------- lazy path-length regularisation (reference :418-444)
        if self.iteration % hp["lazy_generator_regularization"] == 0:
            self._zero()
            n_pl = max(1, int(hp["batch_size_shrink_path_length_regularization"] * batch))
            z = dr.z_pl if dr.z_pl is not None else self._noise(n_pl)
            self.generator_reducer.arm()
            grads = G(input=z, inject_index=dr.inject_pl, noise=dr.noise_pl, return_path_length_grads=True,
                      path_length_noise=dr.pl_image_noise)
            reduce_fn = msg_dist.all_reduce_mean if msg_dist.collectives_active() else None
            pl_loss, path_length = self.path_length_regularization(grads, reduce_fn)
            (hp["w_generator_regularization"] * pl_loss).backward()
            self._step(self.generator_reducer, self.generator_optimizer, "pl")
            self._record(path_length=path_length, loss_path_length_regularization=pl_loss)
        # ---------------- EMA (reference :446)
        if self.step_trace is not None:
            ema_before = {n: p.detach().clone() for n, p in self.generator_ema.named_parameters()}
        misc.exponential_moving_average(model_ema=self.generator_ema, model_train=self.generator)
        if self.step_trace is not None:
            for n, p in self.generator_ema.named_parameters():
                self.step_trace[f"ema.delta.{n}"] = p.detach() - ema_before[n]
