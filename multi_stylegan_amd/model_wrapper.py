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
import os
import random
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import data as msg_data
from . import dist as msg_dist
from . import loss, misc
from . import optim as msg_optim
from .config import generation_hyperparameters
from .u_net_2d_discriminator import generate_cut_mix_augmentation_data, generate_cut_mix_transformation_data


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
    wrong_order_perm: Optional[torch.Tensor] = None       # replaces misc.random_permutation (reference :276)
    cut_mix: Optional[bool] = None                        # replaces the random gate of the CutMix block (:331-332)
    cut_mix_map_aug: Optional[torch.Tensor] = None        # replace the two random binary maps (:337, :357)
    cut_mix_map_reg: Optional[torch.Tensor] = None

    def to(self, device):
        def mv(v):
            if v is None or isinstance(v, (int, bool)):
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
                 trap_weights_map: Optional[torch.Tensor] = None,
                 cut_mix_augmentation_loss: nn.Module = None, cut_mix_regularization_loss: nn.Module = None,
                 device: str = "cuda", lr_generator: float = 2e-4, lr_discriminator: float = 6e-4,
                 bucket_bytes: int = 32 << 20, overlap_communication: bool = True,
                 skip_discriminator_weight_grads_in_generator_step: bool = False,
                 fused_optimizer: Optional[bool] = None, batch_discriminator_passes: bool = True,
                 flat_optimizer_step: bool = True, validation_metrics: Tuple[Any, ...] = ()) -> None:
        self.device = torch.device(device)
        self.validation_metrics = tuple(validation_metrics)     # reference model_wrapper.py:29,69 (validation_metrics.IS / FID / FVD)
        self.best_fvd = float("inf")                            # reference :97
        self.generator = generator.to(self.device)
        self.discriminator = discriminator.to(self.device)
        self.hyperparameters = hyperparameters
        self.generator_loss = generator_loss or loss.NonSaturatingLogisticGeneratorLoss()
        self.discriminator_loss = discriminator_loss or loss.NonSaturatingLogisticDiscriminatorLoss()
        self.discriminator_regularization_loss = discriminator_regularization_loss or loss.R1Regularization()
        self.path_length_regularization = (path_length_regularization or loss.PathLengthRegularization()) \
            .to(self.device)
        self.cut_mix_augmentation_loss = cut_mix_augmentation_loss or loss.NonSaturatingLogisticDiscriminatorLossCutMix()
        self.cut_mix_regularization_loss = cut_mix_regularization_loss or nn.MSELoss(reduction="mean")
        self.trap_weights_map = trap_weights_map
        self.epoch, self.epochs = 0, 1           # the reference's schedule position (model_wrapper.py:138); see train()
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
        def hyper_key(optimizer):
            # parameters that are stepped with different hyper-parameters never share a gradient bucket
            table = {id(p): (g["lr"] if not isinstance(g["lr"], torch.Tensor) else id(g), tuple(g.get("betas", ())),
                             g.get("eps"), g.get("weight_decay", 0))
                     for g in optimizer.param_groups for p in g["params"]}
            return lambda p: table.get(id(p))
        self.generator_reducer = msg_dist.GradBucketReducer(live, bucket_bytes, overlap_communication,
                                                            split_key=hyper_key(self.generator_optimizer))
        self.discriminator_reducer = msg_dist.GradBucketReducer(self.discriminator.parameters(), bucket_bytes,
                                                                overlap_communication,
                                                                split_key=hyper_key(self.discriminator_optimizer))
        # Adam on flat stores (multi_stylegan_amd.optim): one launch per bucket instead of torch's multi-tensor step
        self._flat: Dict[int, msg_optim.FlatAdam] = {}
        if flat_optimizer_step and self.device.type == "cuda":
            for opt, red in ((self.generator_optimizer, self.generator_reducer),
                             (self.discriminator_optimizer, self.discriminator_reducer)):
                if opt.defaults.get("fused") and msg_optim.FlatAdam.supported(opt, red):
                    self._flat[id(opt)] = msg_optim.FlatAdam(opt, red)
            g_flat = self._flat.get(id(self.generator_optimizer))
            if g_flat is not None:
                g_flat.attach_ema(self.generator_ema, self.generator)
        self.skip_d_wgrad = skip_discriminator_weight_grads_in_generator_step
        # only discriminators that know about minibatch groups (ours) can take the concatenated batch
        self.batch_discriminator_passes = batch_discriminator_passes and \
            getattr(discriminator, "supports_minibatch_groups", False)
        # Draws that decide WHICH optimiser steps an iteration contains (the CutMix gate, reference :331-332) must come out
        # the same on every rank: a rank that runs two extra discriminator exchanges while the others start the
        # generator's deadlocks RCCL or mixes up buckets.  Python's global `random` is rank-distinct by design (it also
        # drives the per-rank style-mixing draws), so under data parallelism the gate uses a generator of its own whose
        # seed rank 0 hands to everyone; a single process keeps the reference's `random.random()` stream.
        self._control_rng: Optional[random.Random] = None
        if msg_dist.collectives_active():
            seed = torch.tensor([random.getrandbits(62)], dtype=torch.int64,
                                device=self.device if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.broadcast(seed, src=0)
            self._control_rng = random.Random(int(seed.item()))
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
        flat = self._flat.get(id(optimizer))
        fused = flat is not None or (isinstance(optimizer, torch.optim.Adam) and bool(optimizer.defaults.get("fused")))
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
            if flat is None or not flat.step(coef):
                if flat is not None:
                    flat.release()
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

    def _control_random(self) -> float:
        """``random.random()`` for control flow: identical on every rank (see ``_control_rng``)."""
        local = random.random()          # (the reference's draw: this rank's own stream advances as in a single process)
        return self._control_rng.random() if self._control_rng is not None else local

    def _zero(self) -> None:
        self.discriminator_reducer.zero_grad()
        self.generator_reducer.zero_grad()

    def _top_k(self, top_k, prediction: torch.Tensor, prediction_pixel_wise: torch.Tensor):
        """Reference :392-401.  Returns the kept scalar / pixel-wise predictions and the factor the (mean-reduced)
        generator losses have to be multiplied with.  Single process: torch.topk of the batch, factor 1.  Data parallel:
        the reference ranks the GATHERED batch (DataParallel computes the loss on device 0), so the k best of the global
        batch are kept -- every rank keeps those of them that are its own, and since a rank's mean runs over its own
        k_local kept samples while the ranks' gradients are averaged, its losses are weighted by k_local * world / k."""
        if top_k is None or isinstance(top_k, nn.Identity):
            return prediction, prediction_pixel_wise, 1.0
        if msg_dist.collectives_active() and hasattr(top_k, "calc_v"):
            index, factor = msg_dist.global_top_k(prediction.detach().reshape(-1), top_k.calc_v())
            return prediction.reshape(-1)[index], prediction_pixel_wise[index], factor
        output = top_k(prediction)
        if isinstance(output, tuple):
            return output[0], prediction_pixel_wise[output[1]], 1.0
        return output, prediction_pixel_wise, 1.0

    # ------------------------------------------------------------------------------------------- one iteration
    def train_iteration(self, real_images: torch.Tensor, draws: Optional[Draws] = None,
                        resume_training: bool = False, top_k: Optional[nn.Module] = None) -> None:
        """One pass of the reference's loop body (model_wrapper.py:253-451).  ``self.epoch`` / ``self.epochs`` and
        ``resume_training`` switch on the late-training branches as the reference does: wrongly ordered reals among the
        fakes (:272-277), the trap-region weight map on the pixel-wise losses (:289-291, :404-406), CutMix augmentation
        and consistency regularisation (:331-376); ``top_k`` is the module of loss.py:398-444 (:392-401)."""
        # Backward on THIS thread: the autograd engine's device thread is a second Python thread that every custom backward
        # function of this package has to wake, hand the GIL to and hand results back from -- measured at config 1's shapes
        # (64^2, batch 4, host-bound): 30.6 ms per iteration with the engine's thread, 25.3 ms without (tools/host_profile.py
        # --single-thread).  Same graph, same kernels, same stream; node order in second-order passes no longer depends on
        # two threads' sequence counters.
        with torch.autograd.set_multithreading_enabled(False):
            self._train_iteration(real_images, draws, resume_training, top_k)

    def _train_iteration(self, real_images, draws, resume_training, top_k) -> None:
        hp = self.hyperparameters
        dr = draws or Draws()
        G, D = self.generator, self.discriminator
        self.iteration += 1
        real_images = real_images.to(self.device, non_blocking=True)
        batch = real_images.shape[0]
        late = self.epoch >= hp["wrong_order_start"] * self.epochs or resume_training
        weight = self.trap_weights_map if (hp["trap_weight"] * self.epochs <= self.epoch or resume_training) else None
        # ---------------- discriminator step (reference :258-305)
        self._zero()
        with torch.no_grad():
            z = dr.z_d if dr.z_d is not None else self._noise(batch)
            fake_images = G(input=z, inject_index=dr.inject_d, noise=dr.noise_d)
            if late:            # a few real sequences with their time steps re-ordered count as fakes (:272-277)
                perm = dr.wrong_order_perm if dr.wrong_order_perm is not None else \
                    misc.random_permutation(real_images.shape[2])
                count = max(1, int(hp["batch_factor_wrong_order"] * batch))
                fake_images = torch.cat([fake_images.to(real_images.dtype),
                                         real_images[:count].index_select(2, perm.to(real_images.device))], dim=0)
        self.discriminator_reducer.arm("d")
        if self.batch_discriminator_passes and real_images.shape == fake_images.shape:
            # D(real) and D(fake) of the reference (:272-275) as ONE batch of 2B with per-half minibatch statistics:
            # same result, half the launches, better-filled tiles on the low-resolution layers
            both_in = torch.cat([real_images, fake_images.to(real_images.dtype)])
            both, both_px = D(both_in, minibatch_groups=2)
            (real_prediction, fake_prediction) = both.split(batch)
            (real_prediction_pixel_wise, fake_prediction_pixel_wise) = both_px.split(batch)
            if getattr(D, "augments_in_place", False):
                # ADA rewrites its input batch with the augmented images (reference quirk): R1 and CutMix further down
                # this iteration work on what the discriminator saw
                real_images, fake_images = both_in[:batch], both_in[batch:]
        else:
            real_prediction, real_prediction_pixel_wise = D(real_images, is_real=True, is_cut_mix=False)
            fake_prediction, fake_prediction_pixel_wise = D(fake_images, is_real=False, is_cut_mix=False)
        loss_real, loss_fake = self.discriminator_loss(real_prediction, fake_prediction)
        loss_real_px, loss_fake_px = self.discriminator_loss(real_prediction_pixel_wise, fake_prediction_pixel_wise,
                                                             weight=weight)
        (loss_real + loss_fake + loss_real_px + loss_fake_px).backward()
        self._step(self.discriminator_reducer, self.discriminator_optimizer, "d")
        self._record(loss_discriminator_real=loss_real, loss_discriminator_fake=loss_fake,
                     loss_discriminator_real_pixel_wise=loss_real_px,
                     loss_discriminator_fake_pixel_wise=loss_fake_px)
        # ---------------- lazy R1 (reference :307-329)
        real_for_cut_mix = real_images
        if self.iteration % hp["lazy_discriminator_regularization"] == 0:
            self._zero()
            real_rg = real_images.detach().requires_grad_(True)
            self.discriminator_reducer.arm("r1")
            # (the reference keeps requires_grad on the batch and overwrites the D step's real predictions here, :313-316:
            # a CutMix block in the same iteration mixes THESE predictions)
            real_prediction, real_prediction_pixel_wise = D(real_rg, is_real=False, is_cut_mix=True)
            r1 = self.discriminator_regularization_loss(real_prediction, real_rg, real_prediction_pixel_wise)
            (hp["w_discriminator_regularization_r1"] * r1).backward()
            self._step(self.discriminator_reducer, self.discriminator_optimizer, "r1")
            self._record(loss_discriminator_regularization=r1)
            real_for_cut_mix = real_rg
        # ---------------- CutMix augmentation + consistency regularisation (reference :331-376)
        if dr.cut_mix is not None:
            do_cut_mix = dr.cut_mix
        else:
            do_cut_mix = (self._control_random() <= (0.5 / float(self.epochs)) * float(self.epoch)) or \
                (resume_training and self._control_random() <= 0.5)
        if do_cut_mix:
            w_reg = hp["w_discriminator_regularization"]
            self._zero()
            images, label = generate_cut_mix_augmentation_data(real_for_cut_mix, fake_images, dr.cut_mix_map_aug)
            self.discriminator_reducer.arm("cm_aug")
            _, prediction = D(images, is_cut_mix=True)
            cm_real, cm_fake = self.cut_mix_augmentation_loss(prediction, label)
            (w_reg * (cm_real + cm_fake)).backward()
            self._step(self.discriminator_reducer, self.discriminator_optimizer, "cm_aug")
            self._record(loss_cut_mix_augmentation=cm_real + cm_fake)
            self.discriminator_reducer.zero_grad()
            images, label = generate_cut_mix_transformation_data(
                real_for_cut_mix.detach(), fake_images.detach(), real_prediction_pixel_wise.detach(),
                fake_prediction_pixel_wise.detach(), dr.cut_mix_map_reg)
            self.discriminator_reducer.arm("cm_reg")
            _, prediction = D(images, is_cut_mix=True)
            cm_consistency = self.cut_mix_regularization_loss(prediction, label)
            (w_reg * cm_consistency).backward()
            self._step(self.discriminator_reducer, self.discriminator_optimizer, "cm_reg")
            self._record(loss_cut_mix_regularization=cm_consistency)
        # ---------------- generator step (reference :377-416)
        self._zero()
        z = dr.z_g if dr.z_g is not None else self._noise(batch)
        if self.skip_d_wgrad:
            D.requires_grad_(False)
        self.generator_reducer.arm("g")
        fake_images = G(input=z, inject_index=dr.inject_g, noise=dr.noise_g)
        fake_prediction, fake_prediction_pixel_wise = D(fake_images, is_real=False, is_cut_mix=False)
        fake_prediction, fake_prediction_pixel_wise, factor = self._top_k(top_k, fake_prediction,
                                                                           fake_prediction_pixel_wise)
        loss_g = self.generator_loss(fake_prediction)
        loss_g_px = self.generator_loss(fake_prediction_pixel_wise, weight=weight)
        ((loss_g + loss_g_px) * factor if factor != 1.0 else loss_g + loss_g_px).backward()
        if self.skip_d_wgrad:
            D.requires_grad_(True)
        self._step(self.generator_reducer, self.generator_optimizer, "g")
        self._record(loss_generator=loss_g, loss_generator_pixel_wise=loss_g_px)
        # ---------------- lazy path-length regularisation (reference :418-444)
        if self.iteration % hp["lazy_generator_regularization"] == 0:
            self._zero()
            n_pl = max(1, int(hp["batch_size_shrink_path_length_regularization"] * batch))
            z = dr.z_pl if dr.z_pl is not None else self._noise(n_pl)
            self.generator_reducer.arm("pl")
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
        g_flat = self._flat.get(id(self.generator_optimizer))
        if g_flat is not None:
            g_flat.ema_update(0.999)
        else:
            misc.exponential_moving_average(model_ema=self.generator_ema, model_train=self.generator)
        if self.step_trace is not None:
            for n, p in self.generator_ema.named_parameters():
                self.step_trace[f"ema.delta.{n}"] = p.detach() - ema_before[n]

    # --------------------------------------------------------------------------------------------- epoch loop
    def _gan_training(self, training_dataset, resume_training: bool = False,
                      top_k: Optional[nn.Module] = None) -> None:
        """One epoch: the reference's ``_gan_training`` (model_wrapper.py:245-451) over any iterable of real batches
        ``[B, 2, 3, H, W]``.  Host batches (the reference's pinned-memory DataLoader, train_multi_stylegan.py:60-63, or any
        pageable iterable) go through ``data.DevicePrefetcher``: staged in pinned slots and copied on a side stream one
        iteration ahead, so the step never waits for its input; device-side feeds (``data.SyntheticBatches``) pass through."""
        for real_images in msg_data.prefetch(training_dataset, self.device):
            self.train_iteration(real_images, resume_training=resume_training, top_k=top_k)

    @torch.no_grad()
    def validation(self, training_dataset) -> Dict[str, float]:
        """The reference's ``validation`` (model_wrapper.py:197-243): every metric of ``validation_metrics`` on the EMA
        generator and the training data, scores logged as ``<Class>_bf`` / ``_gfp`` / ``_rfp`` (returned, and kept in the
        wrapper's log), the best bright-field FVD remembered."""
        self.generator_ema.eval()
        scores_out: Dict[str, float] = {}
        for metric in self.validation_metrics:
            scores = metric(generator=self.generator_ema, dataset=training_dataset)
            name = metric.__class__.__name__
            if isinstance(scores, (int, float)):
                scores = (float(scores),)
            for suffix, value in zip(("_bf", "_gfp", "_rfp"), scores):
                scores_out[name + suffix] = float(value)
            if "FVD" in name and self.best_fvd > scores[0]:
                self.best_fvd = float(scores[0])
        self._record(**{k: torch.tensor(v, device=self.device) for k, v in scores_out.items()})
        return scores_out

    def train(self, training_dataset, epochs: int = 20, save_model_after_n_epochs: int = 5,
              resume_training: bool = False, top_k: bool = False, checkpoint_directory: Optional[str] = None,
              on_epoch_end=None, validate_after_n_epochs: int = 10) -> None:
        """The reference's ``train`` (model_wrapper.py:104-195) restricted to the hot path: top-k schedule set-up
        (:115-125), the epoch loop, the metric validation every ``validate_after_n_epochs`` epochs when the wrapper was
        given ``validation_metrics`` (:175-177), a checkpoint in the reference's layout every ``save_model_after_n_epochs``
        epochs (:179-192, written by rank 0).  Sample dumps and the process-title / progress-bar plumbing are outside
        the scope; ``on_epoch_end(wrapper, epoch)`` is the hook for them."""
        steps_per_epoch = len(training_dataset)
        top_k_module: Optional[nn.Module] = None
        if top_k:
            hp = self.hyperparameters
            top_k_module = loss.TopK(starting_iteration=int(hp["top_k_start"] * epochs * steps_per_epoch),
                                     final_iteration=int(hp["top_k_finish"] * epochs * steps_per_epoch))
            if resume_training:
                top_k_module.starting_iteration, top_k_module.final_iteration = 0, 1
        self.epochs = epochs
        for self.epoch in range(epochs):
            self.generator.train()
            self.discriminator.train()
            self._gan_training(training_dataset, resume_training=resume_training, top_k=top_k_module)
            if on_epoch_end is not None:
                on_epoch_end(self, self.epoch)
            if self.validation_metrics and (self.epoch + 1) % validate_after_n_epochs == 0:
                self.validation(training_dataset)
            if checkpoint_directory is not None and (self.epoch + 1) % save_model_after_n_epochs == 0:
                self.save_checkpoint(os.path.join(checkpoint_directory, f"checkpoint_{self.epoch + 1}.pt"))

    # --------------------------------------------------------------------------------------------- checkpoints
    def checkpoint_dict(self, data_parallel_prefix: bool = False) -> Dict[str, Any]:
        """The reference's six checkpoint entries (model_wrapper.py:181-192).  ``path_length_regularization`` carries
        the running path-length mean, which the reference loses (a plain attribute); the extra key
        ``multi_stylegan_amd`` holds what else a resumed run needs (iteration counter, ADA state when the
        discriminator is wrapped).  ``data_parallel_prefix=True`` writes the ``module.`` key prefix of a checkpoint
        saved from ``nn.DataParallel`` models, for consumers that load into wrapped models (scripts/get_gan_samples.py:35)."""
        def sd(module):
            state = module.state_dict()
            return {("module." + k if data_parallel_prefix else k): v for k, v in state.items()}
        extra = {"iteration": self.iteration, "epoch": self.epoch}
        if self._control_rng is not None:
            extra["control_rng"] = self._control_rng.getstate()
        ada = getattr(self.discriminator, "ada_state", None)
        if callable(ada):
            extra["ada"] = ada()
        return {"generator_ema": sd(self.generator_ema), "generator": sd(self.generator),
                "generator_optimizer": self.generator_optimizer.state_dict(),
                "discriminator": sd(self.discriminator),
                "discriminator_optimizer": self.discriminator_optimizer.state_dict(),
                "path_length_regularization": self.path_length_regularization.state_dict(),
                "multi_stylegan_amd": extra}

    def save_checkpoint(self, path: str, data_parallel_prefix: bool = False) -> None:
        """``Logger.save_checkpoint`` of the reference (misc.py:124-130): one ``torch.save`` of the dict; rank 0 only."""
        if msg_dist.collectives_active() and torch.distributed.get_rank() != 0:
            return
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save(self.checkpoint_dict(data_parallel_prefix), path)

    def load_checkpoint(self, checkpoint: Union[str, Dict[str, Any]]) -> None:
        """Load a checkpoint written by ``save_checkpoint`` OR by the reference (train_multi_stylegan.py:74-90): the
        ``module.`` prefix of DataParallel checkpoints and the ``discriminator.`` prefix of an ADA-wrapped reference
        discriminator are stripped (or added, when this wrapper's discriminator is itself ADA-wrapped); a reference
        checkpoint's empty ``path_length_regularization`` entry leaves the running mean at its current value."""
        from . import conv_ops
        if isinstance(checkpoint, str):
            checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)
        self.generator.load_state_dict(_match_keys(checkpoint["generator"], self.generator))
        self.generator_ema.load_state_dict(_match_keys(checkpoint["generator_ema"], self.generator_ema))
        self.discriminator.load_state_dict(_match_keys(checkpoint["discriminator"], self.discriminator))
        self.generator_optimizer.load_state_dict(checkpoint["generator_optimizer"])
        self.discriminator_optimizer.load_state_dict(checkpoint["discriminator_optimizer"])
        for flat in self._flat.values():                 # the loaded moments and step counts go into the flat stores
            flat.adopt()
        pl_state = checkpoint.get("path_length_regularization") or {}
        if "mean_path_length" in pl_state:
            self.path_length_regularization.mean_path_length = \
                pl_state["mean_path_length"].to(self.device, torch.float).reshape(1)
        extra = checkpoint.get("multi_stylegan_amd", {})
        self.iteration = int(extra.get("iteration", self.iteration))
        if self._control_rng is not None and "control_rng" in extra:
            self._control_rng.setstate(extra["control_rng"])
        if "ada" in extra and callable(getattr(self.discriminator, "load_ada_state", None)):
            self.discriminator.load_ada_state(extra["ada"])
        # load_state_dict copies into the parameters in place: kernel-side weight images cached so far are stale
        conv_ops.invalidate_weight_cache()
        # gradients live in the reducers' flat buckets; loading must not have detached them
        self.generator_reducer.zero_grad()
        self.discriminator_reducer.zero_grad()


def _match_keys(state: Dict[str, torch.Tensor], module: nn.Module) -> Dict[str, torch.Tensor]:
    """Rename checkpoint keys to the target module's: drop ``module.`` (nn.DataParallel) path components and a leading
    ``discriminator.`` (the reference's ADA wrapper) that the target does not have; add the latter if only the target
    has it."""
    want = set(module.state_dict().keys())
    wrapped = any(k.startswith("discriminator.") for k in want)
    out = {}
    for key, value in state.items():
        name = ".".join(part for part in key.split(".") if part != "module")
        if name not in want:
            if name.startswith("discriminator.") and not wrapped:
                name = name[len("discriminator."):]
            elif wrapped and "discriminator." + name in want:
                name = "discriminator." + name
        out[name] = value
    return out
