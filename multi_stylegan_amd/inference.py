"""Forward-only sampling path: what the reference does with an EMA checkpoint in
scripts/get_gan_samples.py:30-60 (load ``generator_ema``, ``get_noise(p_mixed_noise=0)``, one forward per sample,
split into the bright-field and the GFP sequence) and with the validation noise in model_wrapper.py:147-174.

On the MI355X a forward pass of the 256^2 generator is ~150 kernel launches of a few microseconds each at batch 1,
so the eager path is bound by the host's launch rate, not by the GPU.  ``GeneratorSampler`` therefore captures the
forward ONCE into a HIP graph (static z / noise / image buffers, weights re-laid before the capture) and replays it
per batch: one graph launch instead of ~150 kernel launches.
"""
from typing import Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import conv_ops, misc


def load_generator_ema(generator: nn.Module, checkpoint: Union[str, Dict]) -> nn.Module:
    """``generator.load_state_dict(torch.load(path)["generator_ema"])`` of get_gan_samples.py:34-35 for a plain (not
    DataParallel-wrapped) generator: the ``module.`` prefix of reference checkpoints is dropped."""
    if isinstance(checkpoint, str):
        checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)
    state = checkpoint["generator_ema"] if "generator_ema" in checkpoint else checkpoint
    state = {".".join(part for part in key.split(".") if part != "module"): value for key, value in state.items()}
    generator.load_state_dict(state)
    conv_ops.invalidate_weight_cache()
    return generator.eval().requires_grad_(False)


class GeneratorSampler:
    """Batched, graph-captured ``generator(z)``.

    ``mixing=False``: one latent per sample (get_gan_samples.py:41, ``p_mixed_noise=0``).  ``mixing=True``: two latents
    and a fixed crossover layer ``inject_index`` (the validation noise of model_wrapper.py:96-99 is a mixed pair; the
    crossover is drawn once here because a captured graph has a fixed structure).
    ``randomize_noise=False`` uses the generator's registered noise buffers (model_wrapper.py:156), ``True`` draws fresh
    per-layer noise inside the graph (graph-safe Philox offsets: every replay gets new noise).
    """

    def __init__(self, generator: nn.Module, batch_size: int = 1, randomize_noise: bool = True, mixing: bool = False,
                 inject_index: Optional[int] = None, use_graph: bool = True, device: Union[str, torch.device] = "cuda"):
        self.generator = generator.to(device).eval().requires_grad_(False)
        self.device = torch.device(device)
        self.batch_size, self.randomize_noise, self.mixing = batch_size, randomize_noise, mixing
        n_latents = len(generator.main_convolutions_1) + 2
        if mixing and inject_index is None:
            inject_index = n_latents // 2
        self.inject_index = inject_index
        ld = generator.latent_dimensions
        self._z = [torch.zeros(batch_size, ld, device=self.device) for _ in range(2 if mixing else 1)]
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._image: Optional[torch.Tensor] = None
        self._stamps: Optional[tuple] = None
        self.use_graph = use_graph and self.device.type == "cuda"

    def _forward(self) -> torch.Tensor:
        z = self._z if self.mixing else self._z[0]
        return self.generator(z, randomize_noise=self.randomize_noise, inject_index=self.inject_index)

    def _weight_stamps(self) -> tuple:
        """What the captured graph depends on besides its static buffers: the kernel-side weight images are built during the
        warm-up, OUTSIDE the graph, and the graph holds pointers to them.  Any weight change (an EMA step on the wrapped
        generator, load_generator_ema, load_checkpoint, an in-place edit) changes these stamps."""
        return tuple(conv_ops._stamp(p) for p in self.generator.parameters())

    def reset(self) -> None:
        """Drop the captured graph; the next call re-captures (and re-lays the weights)."""
        self._graph, self._image, self._stamps = None, None, None

    @torch.no_grad()
    def _capture(self) -> None:
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):                       # warm-up: weight images, FIR factors, pointer tables cached
            for _ in range(2):
                self._forward()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._image = self._forward()
        self._graph = graph
        self._stamps = self._weight_stamps()

    @torch.no_grad()
    def __call__(self, z: Union[torch.Tensor, List[torch.Tensor], None] = None) -> torch.Tensor:
        """Images ``[B, 2, 3, H, W]`` for the given latent(s) (drawn if omitted).  With the graph the result is a
        static buffer that the next call overwrites: clone it to keep it."""
        if z is None:
            z = [torch.randn_like(buf) for buf in self._z]
        z = list(z) if isinstance(z, (list, tuple)) else [z]
        if len(z) != len(self._z) or any(t.shape != buf.shape for t, buf in zip(z, self._z)):
            raise ValueError(f"expected {len(self._z)} latent(s) of shape {tuple(self._z[0].shape)}")
        for buf, t in zip(self._z, z):
            buf.copy_(t, non_blocking=True)
        if not self.use_graph:
            return self._forward()
        if self._graph is not None and self._stamps != self._weight_stamps():
            self.reset()                                    # the weights changed since the capture: stale images in the graph
        if self._graph is None:
            self._capture()
        self._graph.replay()
        return self._image


def split_sequences(sequence: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """get_gan_samples.py:45-55: the bright-field and the GFP sequence of a generated ``[B, 2, T, H, W]`` batch as RGB
    frame stacks ``[B, T, 3, H, W]`` (bright field replicated on the three colour planes, GFP on the green one)."""
    bright_field = sequence[:, 0].unsqueeze(2).expand(-1, -1, 3, -1, -1).contiguous()
    gfp = torch.zeros_like(bright_field)
    gfp[:, :, 1] = sequence[:, 1]
    return bright_field, gfp


@torch.no_grad()
def validation_samples(wrapper, noise=None) -> Dict[str, torch.Tensor]:
    """The four per-epoch sample batches of model_wrapper.py:147-174: EMA and training generator, with the registered
    noise buffers and with fresh noise, on one fixed mixed latent pair (15 samples, model_wrapper.py:96-99)."""
    if noise is None:
        if getattr(wrapper, "validation_input_noise", None) is None:
            wrapper.validation_input_noise = misc.get_noise(batch_size=15, latent_dimension=wrapper.latent_dimensions,
                                                            p_mixed_noise=1.0, device=wrapper.device)
        noise = wrapper.validation_input_noise
    was_training = wrapper.generator.training
    wrapper.generator.eval()
    wrapper.generator_ema.eval()
    out = {"prediction_ema": wrapper.generator_ema(input=noise, randomize_noise=False),
           "prediction_ema_rand": wrapper.generator_ema(input=noise, randomize_noise=True),
           "prediction": wrapper.generator(input=noise, randomize_noise=False),
           "prediction_rand": wrapper.generator(input=noise, randomize_noise=True)}
    wrapper.generator.train(was_training)
    return out
