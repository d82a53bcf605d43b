"""Twin-stream Multi-StyleGAN generator on the gfx950 kernels.

Keeps the reference's nn.Module surface -- class names, constructor and ``forward`` signatures, parameter and
buffer names (multi_stylegan/multi_stylegan_generator.py:15-641) -- so reference checkpoints load, but the data
path is laid out for MI355X:

* feature maps live in HBM channels-last (the 512-channel vector of a pixel is one contiguous 1-2 KiB run, which
  is what both the FIR kernels and the implicit-GEMM contractions want), optionally in bf16;
* noise injection + bias + leaky-ReLU are one kernel pass (``op_static.fused_bias_noise_leaky_relu``);
* the modulated convolution uses one shared weight tensor for the whole batch (``conv_ops.modulated_conv2d``);
* the second stream's main convolutions never reach the image (the reference feeds ``output_1`` to both RGB
  heads, generator.py:184-189); they are executed by default, exactly as the reference does, and can be elided with
  ``elide_dead_branch=True`` -- outputs and gradients are identical either way.
"""
import math
from typing import Any, Dict, Iterable, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from . import conv_ops, equalized_layer
from .op_static import FusedLeakyReLU, blur_bias_act, fused_bias_noise_leaky_relu, rgb_skip, upfirdn2d



def _fir2d(taps, gain=1.0):
    t = torch.tensor(list(taps), dtype=torch.float32)
    k = torch.outer(t, t)
    return k / k.sum() * gain


class Upsample(nn.Module):
    """x2 FIR upsampler of the RGB skip path; the FIR is normalised to sum 1 with no factor^2 gain, so each
    stage attenuates by 1/4 exactly like the reference (generator.py:545-551)."""

    def __init__(self, blur_kernel: List[int] = [1, 3, 3, 1], factor: int = 2) -> None:
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", _fir2d(blur_kernel))
        p = len(blur_kernel) - factor
        self.padding = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return upfirdn2d(input, self.kernel, up=self.factor, pad=self.padding)


class Blur(nn.Module):
    def __init__(self, kernel: List[int], sampling_factor: int = 1, sampling_factor_padding: int = 2,
                 kernel_size: int = 3) -> None:
        super().__init__()
        p = (len(kernel) - sampling_factor_padding) + (kernel_size - 1)
        self.padding = ((p + 1) // 2, p // 2)
        self.register_buffer("kernel", _fir2d(kernel, float(sampling_factor ** 2) if sampling_factor > 1 else 1.0))

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return upfirdn2d(input, self.kernel, pad=self.padding)


class ConstantInput(nn.Module):
    def __init__(self, channel: int, size: Tuple[int, int] = (4, 4)) -> None:
        super().__init__()
        self.input = nn.Parameter(torch.ones(1, channel, size[0], size[1]))

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return self.input.expand(input.shape[0], -1, -1, -1)


class NoiseInjection(nn.Module):
    """Stand-alone form (x + w * noise); inside ``StyledConv2d`` the add is fused into the activation kernel."""

    def __init__(self) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1, dtype=torch.float32))

    @staticmethod
    def draw(input: torch.Tensor) -> torch.Tensor:
        return torch.randn(input.shape[0], 1, input.shape[2], input.shape[3], device=input.device,
                           dtype=torch.float32)

    def forward(self, input: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        if noise is None:
            noise = self.draw(input)
        return input + (self.weight * noise).to(input.dtype)


class _Premodulated:
    """A style that already went through the layer's modulation_mapping (Generator.forward computes all of them in one
    grouped launch, conv_ops.grouped_linear)."""
    __slots__ = ("value",)

    def __init__(self, value: torch.Tensor) -> None:
        self.value = value


def _modulated_style(mc: "ModulatedConv2d", style, bsz: int) -> torch.Tensor:
    if isinstance(style, _Premodulated):
        return style.value.view(bsz, 1, mc.in_channels, 1, 1)
    if mc.modulation_mapping is not None:
        return mc.modulation_mapping(style).view(bsz, 1, mc.in_channels, 1, 1)
    return style


class ModulatedConv2d(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, style_dimension: int,
                 kernel_size: Union[int, Tuple[int, int]] = (3, 3), demodulate: bool = True, upsampling: bool = True,
                 blur_kernel: List[int] = [1, 3, 3, 1], modulation_mapping: bool = True) -> None:
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.demodulate, self.upsampling = demodulate, upsampling
        self.kernel_size = kernel_size if isinstance(kernel_size, tuple) else (kernel_size, kernel_size)
        self.blur = Blur(kernel=blur_kernel, sampling_factor=2, sampling_factor_padding=2,
                         kernel_size=self.kernel_size[0]) if upsampling else None
        self.scale = math.sqrt(2) / math.sqrt(in_channels * self.kernel_size[0] * self.kernel_size[1])
        self.weight = nn.Parameter(torch.randn(1, out_channels, in_channels, *self.kernel_size))
        self.modulation_mapping = None
        if modulation_mapping:
            self.modulation_mapping = equalized_layer.EqualizedLinear(style_dimension, in_channels, bias=True)
            self.modulation_mapping.bias.data.fill_(1.0)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, " \
               f"demodulate={self.demodulate}, upsampling={self.upsampling}"

    def forward(self, input: torch.Tensor, style: torch.Tensor, skip_blur: bool = False):
        bsz, feats = input.shape[:2]
        assert feats == self.in_channels, f"Expect input feature shape of {self.in_channels} but get {feats}."
        modulated_style = _modulated_style(self, style, bsz)
        output = conv_ops.modulated_conv2d(input, self.weight, modulated_style.reshape(bsz, self.in_channels),
                                           demodulate=self.demodulate, upsample=self.upsampling)
        if self.upsampling and not skip_blur:
            output = self.blur(output)
        if self.modulation_mapping is not None:
            return output, modulated_style
        return output


class StyledConv2d(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: Union[int, Tuple[int, int]],
                 style_dimension: int, demodulate: bool = True, upsampling: bool = False,
                 blur_kernel: List[int] = [1, 3, 3, 1], modulation_mapping: bool = True) -> None:
        super().__init__()
        self.modulation_mapping = modulation_mapping
        self.modulated_convolution = ModulatedConv2d(in_channels, out_channels, style_dimension, kernel_size,
                                                     demodulate, upsampling, blur_kernel, modulation_mapping)
        self.noise_injection = NoiseInjection()
        self.activation = FusedLeakyReLU(out_channels)

    def forward(self, input: torch.Tensor, style: torch.Tensor, noise: torch.Tensor = None, head_slot=None, act_handle=None,
                input_act=None):
        mc = self.modulated_convolution
        if not mc.upsampling and input.is_cuda:
            # conv -> noise -> bias -> leaky ReLU in one launch (the upsampling layers blur in between: two passes)
            bsz = input.shape[0]
            style_out = _modulated_style(mc, style, bsz)
            if noise is None:
                noise = torch.randn(bsz, 1, input.shape[2], input.shape[3], device=input.device, dtype=torch.float32)
            output = conv_ops.modulated_conv2d_bias_act(
                input, mc.weight, style_out.reshape(bsz, mc.in_channels), mc.demodulate, self.activation.bias, noise,
                self.noise_injection.weight, self.activation.negative_slope, self.activation.scale, head_slot=head_slot,
                input_act=input_act)
            return (output, style_out) if self.modulation_mapping else output
        if mc.upsampling and input.is_cuda:
            # transposed conv -> [blur -> noise -> bias -> leaky ReLU] with the bracket in one launch
            result = mc(input, style, skip_blur=True)
            output, style_out = result if self.modulation_mapping else (result, None)
            p0, p1 = mc.blur.padding
            oh, ow = output.shape[2] + p0 + p1 - 3, output.shape[3] + p0 + p1 - 3
            if noise is None:
                noise = torch.randn(output.shape[0], 1, oh, ow, device=input.device, dtype=torch.float32)
            output = blur_bias_act(output, mc.blur.kernel, mc.blur.padding, self.activation.bias, noise,
                                   self.noise_injection.weight, self.activation.negative_slope, self.activation.scale,
                                   act_handle=act_handle)
            return (output, style_out) if self.modulation_mapping else output
        result = self.modulated_convolution(input, style)
        output, style_out = result if self.modulation_mapping else (result, None)
        if noise is None:
            noise = NoiseInjection.draw(output)
        output = fused_bias_noise_leaky_relu(output, self.activation.bias, noise, self.noise_injection.weight,
                                             self.activation.negative_slope, self.activation.scale)
        if self.modulation_mapping:
            return output, style_out
        return output


def _merge_rgb(conv: torch.Tensor, bias: torch.Tensor, skip: Optional[torch.Tensor], upsampling: nn.Module) -> torch.Tensor:
    """conv.float() + bias + upsampling(skip) (multi_stylegan_generator.py:519-523 of the reference) -- one launch
    (op_static.rgb_skip) when the upsampler is the reference's x2 / 4-tap FIR, the composition of ops otherwise."""
    fir = getattr(upsampling, "kernel", None) if skip is not None else None
    if skip is None or isinstance(upsampling, Upsample):
        factor, pad = (upsampling.factor, upsampling.padding) if skip is not None else (2, (2, 1))
        if rgb_skip.supported(conv, skip, fir, factor, pad):
            return rgb_skip.rgb_skip_merge(conv, bias.reshape(-1).float(), skip, fir)
    output = conv.float().contiguous() + bias
    if skip is not None:
        output = output + upsampling(skip)
    return output


class OutputBlock(nn.Module):
    """1x1 modulated conv to the 3 time-step planes (no demodulation) + scalar bias + FIR-upsampled skip.
    The RGB path is small and is kept in fp32 / NCHW."""

    def __init__(self, in_channels: int, style_dimension: int, out_channels: int = 1, upsampling: bool = False,
                 blur_kernel: List[int] = [1, 3, 3, 1], modulation_mapping: bool = True) -> None:
        super().__init__()
        self.modulation_mapping = modulation_mapping
        self.upsampling = Upsample(blur_kernel=blur_kernel, factor=2) if upsampling else nn.Identity()
        self.modulated_convolution = ModulatedConv2d(in_channels, out_channels, style_dimension, (1, 1),
                                                     demodulate=False, upsampling=False,
                                                     modulation_mapping=modulation_mapping)
        self.bias = nn.Parameter(torch.zeros(1, 1, 1, 1, dtype=torch.float32))

    def forward(self, input: torch.Tensor, style: torch.Tensor, skip: torch.Tensor = None):
        result = self.modulated_convolution(input, style)
        output, style_out = result if self.modulation_mapping else (result, None)
        output = _merge_rgb(output, self.bias.expand(1, output.shape[1], 1, 1), skip, self.upsampling)
        if self.modulation_mapping:
            return output, style_out
        return output


class StyleMapping(nn.Module):
    def __init__(self, latent_dimensions: int = 512, depth: int = 8) -> None:
        super().__init__()
        layers: List[nn.Module] = [equalized_layer.PixelwiseNormalization()]
        for _ in range(depth):
            layers += [equalized_layer.EqualizedLinear(latent_dimensions, latent_dimensions, bias=False),
                       FusedLeakyReLU(latent_dimensions)]
        self.layers = nn.Sequential(*layers)

    def forward(self, noise: torch.Tensor) -> torch.Tensor:
        return self.layers(noise)


class Generator(nn.Module):
    def __init__(self, config: Dict[str, Any]) -> None:
        super().__init__()
        channels = [int(c // config["channel_factor"]) for c in config["channels"]]
        self.out_channels: int = 3
        self.latent_dimensions: int = config["latent_dimensions"]
        self.starting_resolution: Tuple[int, int] = config["starting_resolution"]
        ld, c0 = self.latent_dimensions, channels[0]
        self.style_mapping = StyleMapping(latent_dimensions=ld, depth=config["depth_style_mapping"])
        self.constant_input_1 = ConstantInput(channel=c0, size=self.starting_resolution)
        self.constant_input_2 = ConstantInput(channel=c0, size=self.starting_resolution)
        self.starting_convolution_1 = StyledConv2d(c0, c0, (3, 3), ld)
        self.starting_convolution_2 = StyledConv2d(c0, c0, (3, 3), ld, modulation_mapping=False)
        self.starting_output_block_1 = OutputBlock(c0, ld, self.out_channels)
        self.starting_output_block_2 = OutputBlock(c0, ld, self.out_channels, modulation_mapping=False)
        self.main_convolutions_1, self.output_blocks_1 = nn.ModuleList(), nn.ModuleList()
        self.main_convolutions_2, self.output_blocks_2 = nn.ModuleList(), nn.ModuleList()
        for c_in, c_out in zip(channels[:-1], channels[1:]):
            for convs, heads, mapped in ((self.main_convolutions_1, self.output_blocks_1, True),
                                         (self.main_convolutions_2, self.output_blocks_2, False)):
                convs.append(StyledConv2d(c_in, c_out, (2, 2), ld, upsampling=True, modulation_mapping=mapped))
                convs.append(StyledConv2d(c_out, c_out, (3, 3), ld, modulation_mapping=mapped))
                heads.append(OutputBlock(c_out, ld, self.out_channels, upsampling=True, modulation_mapping=mapped))
        self.noises = nn.Module()
        self.noises.register_buffer("noise_start", torch.randn(1, 1, *self.starting_resolution))
        for level in range(len(channels) - 1):
            res = 2 ** (level + 3)
            self.noises.register_buffer(f"noise_{2 * level}", torch.randn(1, 1, res, res))
            self.noises.register_buffer(f"noise_{2 * level + 1}", torch.randn(1, 1, res, res))
        # MI355X-side knobs (not part of the reference surface; defaults reproduce it)
        self.compute_dtype = torch.float32
        self.elide_dead_branch = False

    # ------------------------------------------------------------------ reference API
    def get_parameters(self, lr_main: float = 1e-03, lr_style: float = 1e-05) -> Iterable:
        order = ["constant_input_1", "starting_convolution_1", "starting_output_block_1", "main_convolutions_1",
                 "output_blocks_1", "constant_input_2", "starting_convolution_2", "starting_output_block_2",
                 "main_convolutions_2", "output_blocks_2"]
        groups = [{"params": getattr(self, name).parameters(), "lr": lr_main} for name in order]
        groups.append({"params": self.style_mapping.parameters(), "lr": lr_style})
        return groups

    def live_parameters(self) -> List[nn.Parameter]:
        """Parameters that can ever receive a gradient (everything but the dead second-stream main convs)."""
        return [p for n, p in self.named_parameters() if not n.startswith("main_convolutions_2.")]

    def _latent(self, input, inject_index, input_is_latent):
        n = len(self.main_convolutions_1) + 2
        if input_is_latent:
            if input.ndim < 3:
                return input.unsqueeze(1).repeat(1, n, 1)
            if input.shape[1] != n:
                return input.repeat(1, n, 1)
            return input
        if isinstance(input, (list, tuple)):
            if len(input) > 1 and all(z.shape == input[0].shape for z in input):
                # the mapping network is row-wise: both latent draws of a style-mixing pair go through it as one batch
                styles = self.style_mapping(torch.cat(list(input), dim=0)).chunk(len(input), dim=0)
            else:
                styles = [self.style_mapping(z) for z in input]
            if inject_index is None:
                inject_index = np.random.randint(1, n - 1)
            return torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                              styles[1].unsqueeze(1).repeat(1, n - inject_index, 1)], dim=1)
        return self.style_mapping(input).unsqueeze(1).repeat(1, n, 1)

    def _style_groups(self):
        """(modulated conv, latent slot) of every layer that owns a modulation_mapping, in execution order."""
        groups = [(self.starting_convolution_1.modulated_convolution, 0),
                  (self.starting_output_block_1.modulated_convolution, 1)]
        for i in range(len(self.main_convolutions_1) // 2):
            groups += [(self.main_convolutions_1[2 * i].modulated_convolution, 2 * i + 1),
                       (self.main_convolutions_1[2 * i + 1].modulated_convolution, 2 * i + 2),
                       (self.output_blocks_1[i].modulated_convolution, 2 * i + 3)]
        return groups

    def _all_styles(self, latent: torch.Tensor):
        """All style affines (multi_stylegan_generator.py:379-382, one EqualizedLinear per styled layer) in ONE grouped
        launch: they only depend on the latent.  None when the layers are not uniform (then each layer maps its own)."""
        if not (latent.is_cuda and latent.dtype == torch.float32):
            return None
        groups = self._style_groups()
        maps = [mc.modulation_mapping for mc, _ in groups]
        first = maps[0]
        if any(m is None or m.bias is None or m.weight.shape != first.weight.shape for m in maps):
            return None
        out = conv_ops.grouped_linear(latent, [slot for _, slot in groups], [m.weight for m in maps],
                                      [m.bias for m in maps], first.scale, first.scale_bias)
        return out.unbind(0)

    def _heads_pairable(self, features: torch.Tensor) -> bool:
        if not features.is_cuda:
            return False
        for h1, h2 in zip(self.output_blocks_1, self.output_blocks_2):
            m1, m2 = h1.modulated_convolution, h2.modulated_convolution
            if m1.modulation_mapping is None or m2.modulation_mapping is not None or m1.demodulate or m2.demodulate \
                    or m1.weight.shape != m2.weight.shape:
                return False
        return True

    @staticmethod
    def _paired_heads(head1: "OutputBlock", head2: "OutputBlock", features: torch.Tensor, latent_w, skip: torch.Tensor,
                      head_slot=None):
        """output_blocks_1[i](features, w) and output_blocks_2[i](features, style_1) as ONE block on a 2 x 3-channel
        map: the reference's second head reads stream 1's features with stream 1's modulated style
        (multi_stylegan_generator.py:184-189) and neither head demodulates, so the two 1x1 modulated convs are one
        contraction with the weights stacked along the output channels -- the 512-channel map is read once, and
        backward gets its gradient from one 6-channel data-gradient conv instead of two 3-channel ones plus an add over
        the whole map.  The skip path (bias, FIR upsampling of the previous level, sum) runs once on the stacked map
        too; `skip` is [B, 2*3, h, w], heads stacked along the channels."""
        mc1, mc2 = head1.modulated_convolution, head2.modulated_convolution
        bsz, o1, o2 = features.shape[0], mc1.out_channels, mc2.out_channels
        style = _modulated_style(mc1, latent_w, bsz)
        both = conv_ops.modulated_conv2d(features, torch.cat([mc1.weight, mc2.weight], dim=1),
                                         style.reshape(bsz, mc1.in_channels), demodulate=False, upsample=False,
                                         head_slot=head_slot)
        bias = torch.cat([head1.bias.expand(1, o1, 1, 1), head2.bias.expand(1, o2, 1, 1)], dim=1)
        return _merge_rgb(both, bias, skip, head1.upsampling), style

    def forward(self, input: Union[List[torch.Tensor], torch.Tensor], return_main_style_vectors: bool = False,
                noise: Optional[List[torch.Tensor]] = None, randomize_noise: bool = True,
                inject_index: Optional[int] = None, input_is_latent: bool = False,
                return_path_length_grads: bool = False, path_length_noise: Optional[torch.Tensor] = None):
        return self._forward(input, return_main_style_vectors, noise, randomize_noise, inject_index, input_is_latent,
                             bool(return_path_length_grads), path_length_noise)

    def _draw_layer_noise(self, latent):
        """Every NoiseInjection's own N(0, 1) map (the reference draws one per layer and stream when no noise is passed,
        multi_stylegan_generator.py:288-292) out of ONE normal draw per forward instead of ~27 launches of a few microseconds'
        work each: -> (start 1, start 2, [stream-1 layers], [stream-2 layers]) as views of one buffer, or None off the GPU."""
        if not latent.is_cuda:
            return None
        b = latent.shape[0]
        res0 = tuple(self.constant_input_1.input.shape[2:])
        shapes = [res0, res0]
        h, w = res0
        for i in range(len(self.main_convolutions_1)):
            if i % 2 == 0:
                h, w = 2 * h, 2 * w
            shapes += [(h, w), (h, w)]                                    # (stream 1, stream 2)
        sizes = [b * hh * ww for hh, ww in shapes]
        flat = torch.randn(sum(sizes), device=latent.device, dtype=torch.float32)
        maps = [t.view(b, 1, hh, ww) for t, (hh, ww) in zip(flat.split(sizes), shapes)]
        return maps[0], maps[1], maps[2::2], maps[3::2]

    def _forward(self, input, return_main_style_vectors, noise, randomize_noise, inject_index, input_is_latent,
                 return_path_length_grads, path_length_noise):
        latent = self._latent(input, inject_index, input_is_latent)
        n_main = len(self.main_convolutions_1)
        layer_noise_2 = None
        if noise is None:
            if randomize_noise:
                noise_start, layer_noise = None, [None] * n_main
                drawn = self._draw_layer_noise(latent)
                if drawn is not None:
                    noise_start, noise_start_2, layer_noise, layer_noise_2 = drawn
            else:
                noise_start = self.noises.noise_start
                layer_noise = [getattr(self.noises, f"noise_{i}") for i in range(n_main)]
        else:
            noise_start, layer_noise = noise[0], list(noise[1:])
        dt = self.compute_dtype
        # (the path-length pass too since round 5: the grouped op's latent gradient is a differentiable node of its own,
        #  conv_ops._GroupedLinD -- before, that pass ran the 20 affines layer by layer through three orders of autograd)
        pre = self._all_styles(latent)
        w_of = (lambda group, slot: _Premodulated(pre[group])) if pre is not None else \
            (lambda group, slot: latent[:, slot])
        out1 = conv_ops.to_compute_layout(self.constant_input_1(latent), dt)
        out2 = conv_ops.to_compute_layout(self.constant_input_2(latent), dt)
        out1, style = self.starting_convolution_1(out1, w_of(0, 0), noise=noise_start)
        out2 = self.starting_convolution_2(out2, style, noise=noise_start if layer_noise_2 is None else noise_start_2)
        skip1, style = self.starting_output_block_1(out1, w_of(1, 1))
        skip2 = self.starting_output_block_2(out2, style)
        run_stream2 = not self.elide_dead_branch
        paired = n_main >= 2 and skip1.shape == skip2.shape and self._heads_pairable(out1)
        skip = torch.cat([skip1, skip2], dim=1) if paired else None
        for i in range(n_main // 2):
            # (the upsampling layer's output feeds the level's 3x3 layer and nothing else: its activation backward rides in that
            #  layer's data-gradient epilogue, conv_ops.ActHandle)
            handle = conv_ops.ActHandle() if out1.is_cuda and torch.is_grad_enabled() and not return_path_length_grads else None
            out1, style = self.main_convolutions_1[2 * i](out1, w_of(2 + 3 * i, 2 * i + 1),
                                                           noise=layer_noise[2 * i], act_handle=handle)
            if run_stream2:
                out2 = self.main_convolutions_2[2 * i](out2, style, noise=(layer_noise_2 or layer_noise)[2 * i])
            # (the level's image heads read out1 and nothing else does but the next level: their data gradient is formed
            #  inside out1's activation backward instead of being written as a map and summed, conv_ops.HeadGradSlot)
            slot = conv_ops.HeadGradSlot() if paired and out1.dtype == torch.bfloat16 and torch.is_grad_enabled() and \
                not return_path_length_grads else None
            out1, style = self.main_convolutions_1[2 * i + 1](out1, w_of(3 + 3 * i, 2 * i + 2),
                                                               noise=layer_noise[2 * i + 1], head_slot=slot, input_act=handle)
            if run_stream2:
                out2 = self.main_convolutions_2[2 * i + 1](out2, style, noise=(layer_noise_2 or layer_noise)[2 * i + 1])
            if paired:
                skip, style = self._paired_heads(self.output_blocks_1[i], self.output_blocks_2[i], out1,
                                                 w_of(4 + 3 * i, 2 * i + 3), skip, head_slot=slot)
            else:
                skip1, style = self.output_blocks_1[i](out1, w_of(4 + 3 * i, 2 * i + 3), skip=skip1)
                skip2 = self.output_blocks_2[i](out1, style, skip=skip2)     # reads stream 1, as the reference does
        if paired:      # [B, 2*3, H, W] -> [B, 2, 3, H, W]: a view
            image = skip.view(skip.shape[0], 2, skip.shape[1] // 2, *skip.shape[2:])
        else:
            image = torch.stack([skip1, skip2], dim=1)
        if return_path_length_grads:
            if path_length_noise is None:
                path_length_noise = torch.randn(image.shape, device=image.device, dtype=torch.float32)
            pl_noise = path_length_noise / math.sqrt(image.shape[2] * image.shape[3] * image.shape[4])
            return torch.autograd.grad((image * pl_noise).sum(), latent, create_graph=True, retain_graph=True)[0]
        if return_main_style_vectors:
            return image, latent
        return image
