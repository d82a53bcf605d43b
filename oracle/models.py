"""Oracle restatement of the generator / discriminator wiring (CPU, plain torch).

Module and parameter names reproduce the reference's ``state_dict`` keys so a
reference checkpoint loads here unchanged.  The math is
delegated to ``oracle.ops``.
"""
import math
from typing import List, Optional, Sequence, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

GENERATOR_CONFIG = {  # multi_stylegan/config.py:16-27
    "channels": (512,) * 7, "channel_factor": 1, "latent_dimensions": 512,
    "depth_style_mapping": 8, "starting_resolution": (4, 4)}
DISCRIMINATOR_CONFIG = {  # multi_stylegan/config.py:6-13
    "encoder_channels": ((3, 128), (128, 256), (256, 384), (384, 768), (768, 1024)),
    "decoder_channels": ((1024, 768), (768, 384), (384, 256), (256, 128)), "fft": False}


# ---------------------------------------------------------------- layers ---
class EqualizedLinear(nn.Module):
    """equalized_layer.py:210-254."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def forward(self, x):
        return ops.equalized_linear(x, self.weight, self.bias)


class EqualizedConv2d(nn.Module):
    """equalized_layer.py:9-74."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True):
        super().__init__()
        k = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.stride = stride if isinstance(stride, int) else stride[0]
        self.padding = padding if isinstance(padding, int) else padding[0]
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, *k))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def forward(self, x):
        return ops.equalized_conv2d(x, self.weight, self.bias, self.stride, self.padding)


class PixelwiseNormalization(nn.Module):
    def forward(self, x):
        return ops.pixel_norm(x)


class FusedLeakyReLU(nn.Module):
    """op_static/fused_act.py:76-85 (module default scale 1.0, quirk Q4)."""

    def __init__(self, channel, negative_slope=0.2, scale=1.0):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope, self.scale = negative_slope, scale

    def forward(self, x):
        return ops.fused_leaky_relu(x, self.bias, self.negative_slope, self.scale)


class Upsample(nn.Module):
    """FIR x2 upsampler WITHOUT the factor^2 gain (quirk Q3); multi_stylegan_generator.py:529-575."""

    def __init__(self, blur_kernel=(1, 3, 3, 1), factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", ops.make_fir(blur_kernel))
        p = len(blur_kernel) - factor
        self.padding = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, x):
        return ops.upfirdn2d(x, self.kernel, up=self.factor, pad=self.padding)


class Blur(nn.Module):
    """multi_stylegan_generator.py:578-641 / u_net_2d_discriminator.py:269-332."""

    def __init__(self, kernel=(1, 3, 3, 1), sampling_factor=1, sampling_factor_padding=2, kernel_size=3):
        super().__init__()
        p = (len(kernel) - sampling_factor_padding) + (kernel_size - 1)
        self.padding = ((p + 1) // 2, p // 2)
        self.register_buffer("kernel", ops.make_fir(kernel, float(sampling_factor ** 2)))

    def forward(self, x):
        return ops.upfirdn2d(x, self.kernel, pad=self.padding)


# ------------------------------------------------------------- generator ---
class ModulatedConv2d(nn.Module):
    """multi_stylegan_generator.py:295-414."""

    def __init__(self, in_channels, out_channels, style_dimension, kernel_size=(3, 3), demodulate=True,
                 upsampling=True, blur_kernel=(1, 3, 3, 1), modulation_mapping=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.demodulate, self.upsampling = demodulate, upsampling
        self.blur = Blur(blur_kernel, 2, 2, kernel_size[0]) if upsampling else None
        self.weight = nn.Parameter(torch.randn(1, out_channels, in_channels, *kernel_size))
        self.modulation_mapping = None
        if modulation_mapping:
            self.modulation_mapping = EqualizedLinear(style_dimension, in_channels, bias=True)
            self.modulation_mapping.bias.data.fill_(1.0)

    def forward(self, x, style):
        bsz = x.shape[0]
        s = style
        if self.modulation_mapping is not None:
            s = self.modulation_mapping(style).view(bsz, 1, self.in_channels, 1, 1)
        y = ops.modulated_conv2d(
            x, self.weight, s.reshape(bsz, self.in_channels), demodulate=self.demodulate,
            upsample=self.upsampling, blur_fir=None if self.blur is None else self.blur.kernel,
            blur_pad=(2, 1) if self.blur is None else self.blur.padding)
        return (y, s) if self.modulation_mapping is not None else y


class NoiseInjection(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, x, noise=None):
        if noise is None:
            noise = torch.randn(x.shape[0], 1, x.shape[2], x.shape[3], device=x.device, dtype=torch.float32)
        return ops.noise_injection(x, self.weight, noise)


class StyledConv2d(nn.Module):
    """multi_stylegan_generator.py:417-469."""

    def __init__(self, in_channels, out_channels, kernel_size, style_dimension, demodulate=True,
                 upsampling=False, modulation_mapping=True):
        super().__init__()
        self.modulation_mapping = modulation_mapping
        self.modulated_convolution = ModulatedConv2d(in_channels, out_channels, style_dimension, kernel_size,
                                                     demodulate, upsampling,
                                                     modulation_mapping=modulation_mapping)
        self.noise_injection = NoiseInjection()
        self.activation = FusedLeakyReLU(out_channels)

    def forward(self, x, style, noise=None):
        res = self.modulated_convolution(x, style)
        y, s = res if self.modulation_mapping else (res, None)
        y = self.activation(self.noise_injection(y, noise))
        return (y, s) if self.modulation_mapping else y


class OutputBlock(nn.Module):
    """multi_stylegan_generator.py:472-526."""

    def __init__(self, in_channels, style_dimension, out_channels=1, upsampling=False, modulation_mapping=True):
        super().__init__()
        self.modulation_mapping = modulation_mapping
        self.upsampling = Upsample() if upsampling else nn.Identity()
        self.modulated_convolution = ModulatedConv2d(in_channels, out_channels, style_dimension, (1, 1),
                                                     demodulate=False, upsampling=False,
                                                     modulation_mapping=modulation_mapping)
        self.bias = nn.Parameter(torch.zeros(1, 1, 1, 1))

    def forward(self, x, style, skip=None):
        res = self.modulated_convolution(x, style)
        y, s = res if self.modulation_mapping else (res, None)
        y = y + self.bias
        if skip is not None:
            y = y + self.upsampling(skip)
        return (y, s) if self.modulation_mapping else y


class StyleMapping(nn.Module):
    def __init__(self, latent_dimensions=512, depth=8):
        super().__init__()
        mods: List[nn.Module] = [PixelwiseNormalization()]
        for _ in range(depth):
            mods += [EqualizedLinear(latent_dimensions, latent_dimensions, bias=False),
                     FusedLeakyReLU(latent_dimensions)]
        self.layers = nn.Sequential(*mods)

    def forward(self, z):
        return self.layers(z)


class ConstantInput(nn.Module):
    def __init__(self, channel, size=(4, 4)):
        super().__init__()
        self.input = nn.Parameter(torch.ones(1, channel, *size))

    def forward(self, latent):
        return self.input.expand(latent.shape[0], -1, -1, -1)


class Generator(nn.Module):
    """multi_stylegan_generator.py:15-205 (twin-stream generator, quirk Q1 kept)."""

    def __init__(self, config=GENERATOR_CONFIG):
        super().__init__()
        ch = [int(c // config["channel_factor"]) for c in config["channels"]]
        ld = self.latent_dimensions = config["latent_dimensions"]
        res0 = self.starting_resolution = config["starting_resolution"]
        self.out_channels = 3
        self.style_mapping = StyleMapping(ld, config["depth_style_mapping"])
        self.constant_input_1 = ConstantInput(ch[0], res0)
        self.constant_input_2 = ConstantInput(ch[0], res0)
        self.starting_convolution_1 = StyledConv2d(ch[0], ch[0], (3, 3), ld)
        self.starting_convolution_2 = StyledConv2d(ch[0], ch[0], (3, 3), ld, modulation_mapping=False)
        self.starting_output_block_1 = OutputBlock(ch[0], ld, 3)
        self.starting_output_block_2 = OutputBlock(ch[0], ld, 3, modulation_mapping=False)
        self.main_convolutions_1, self.output_blocks_1 = nn.ModuleList(), nn.ModuleList()
        self.main_convolutions_2, self.output_blocks_2 = nn.ModuleList(), nn.ModuleList()
        for a, b in zip(ch[:-1], ch[1:]):
            for convs, blocks, mm in ((self.main_convolutions_1, self.output_blocks_1, True),
                                      (self.main_convolutions_2, self.output_blocks_2, False)):
                convs.append(StyledConv2d(a, b, (2, 2), ld, upsampling=True, modulation_mapping=mm))
                convs.append(StyledConv2d(b, b, (3, 3), ld, modulation_mapping=mm))
                blocks.append(OutputBlock(b, ld, 3, upsampling=True, modulation_mapping=mm))
        self.noises = nn.Module()
        self.noises.register_buffer("noise_start", torch.randn(1, 1, *res0))
        for i in range(len(ch) - 1):
            r = 2 ** (i + 3)
            self.noises.register_buffer(f"noise_{2 * i}", torch.randn(1, 1, r, r))
            self.noises.register_buffer(f"noise_{2 * i + 1}", torch.randn(1, 1, r, r))

    @property
    def num_latents(self):
        return len(self.main_convolutions_1) + 2

    def get_parameters(self, lr_main=1e-3, lr_style=1e-5):
        names = ["constant_input_1", "starting_convolution_1", "starting_output_block_1", "main_convolutions_1",
                 "output_blocks_1", "constant_input_2", "starting_convolution_2", "starting_output_block_2",
                 "main_convolutions_2", "output_blocks_2"]
        groups = [{"params": getattr(self, n).parameters(), "lr": lr_main} for n in names]
        return groups + [{"params": self.style_mapping.parameters(), "lr": lr_style}]

    def make_latent(self, z, inject_index=None, input_is_latent=False):
        """:134-172: map z (tensor or list of two) to the [B, n, D] latent stack."""
        n = self.num_latents
        if input_is_latent:
            if z.ndim < 3:
                return z.unsqueeze(1).repeat(1, n, 1)
            return z if z.shape[1] == n else z.repeat(1, n, 1)
        if isinstance(z, (list, tuple)):
            w = [self.style_mapping(t) for t in z]
            if inject_index is None:
                inject_index = np.random.randint(1, n - 1)
            return torch.cat([w[0].unsqueeze(1).repeat(1, inject_index, 1),
                              w[1].unsqueeze(1).repeat(1, n - inject_index, 1)], dim=1)
        return self.style_mapping(z).unsqueeze(1).repeat(1, n, 1)

    def forward(self, input, return_main_style_vectors=False, noise=None, randomize_noise=True,
                inject_index=None, input_is_latent=False, return_path_length_grads=False):
        latent = self.make_latent(input, inject_index, input_is_latent)
        n_main = len(self.main_convolutions_1)
        if noise is None:
            if randomize_noise:
                n0, layer_noise = None, [None] * n_main
            else:
                n0 = self.noises.noise_start
                layer_noise = [getattr(self.noises, f"noise_{i}") for i in range(n_main)]
        else:
            n0, layer_noise = noise[0], list(noise[1:])
        o1, s = self.starting_convolution_1(self.constant_input_1(latent), latent[:, 0], noise=n0)
        o2 = self.starting_convolution_2(self.constant_input_2(latent), s, noise=n0)
        skip1, s = self.starting_output_block_1(o1, latent[:, 1])
        skip2 = self.starting_output_block_2(o2, s)
        for i in range(n_main // 2):
            o1, s = self.main_convolutions_1[2 * i](o1, latent[:, 2 * i + 1], noise=layer_noise[2 * i])
            o2 = self.main_convolutions_2[2 * i](o2, s, noise=layer_noise[2 * i])
            o1, s = self.main_convolutions_1[2 * i + 1](o1, latent[:, 2 * i + 2], noise=layer_noise[2 * i + 1])
            o2 = self.main_convolutions_2[2 * i + 1](o2, s, noise=layer_noise[2 * i + 1])
            skip1, s = self.output_blocks_1[i](o1, latent[:, 2 * i + 3], skip=skip1)
            skip2 = self.output_blocks_2[i](o1, s, skip=skip2)      # o1, not o2: quirk Q1 (:189)
        image = torch.stack([skip1, skip2], dim=1)
        if return_path_length_grads:                                # :193-200, quirk Q9
            pl_noise = torch.randn(image.shape, device=image.device, dtype=torch.float32, requires_grad=True) \
                / math.sqrt(image.shape[2] * image.shape[3] * image.shape[4])
            return torch.autograd.grad((image * pl_noise).sum(), latent, create_graph=True, retain_graph=True)[0]
        return (image, latent) if return_main_style_vectors else image


# --------------------------------------------------------- discriminator ---
class MinibatchStdDev(nn.Module):
    def forward(self, x):
        return ops.minibatch_stddev(x)


class ResNetBlock(nn.Module):
    """u_net_2d_discriminator.py:143-186."""

    def __init__(self, in_channels, out_channels, mini_batch_std_dev=False):
        super().__init__()
        self.mini_batch_std_dev = MinibatchStdDev() if mini_batch_std_dev else nn.Identity()
        self.main_mapping = nn.Sequential(
            EqualizedConv2d(in_channels + int(mini_batch_std_dev), out_channels, 3, 1, 1, bias=False),
            FusedLeakyReLU(out_channels),
            EqualizedConv2d(out_channels, out_channels, 3, 1, 1, bias=False),
            FusedLeakyReLU(out_channels))
        self.residual_mapping = EqualizedConv2d(in_channels, out_channels, 1, 1, 0, bias=False) \
            if in_channels != out_channels else nn.Identity()

    def forward(self, x):
        y = self.main_mapping(self.mini_batch_std_dev(x))
        return (y + self.residual_mapping(x)) / math.sqrt(2)


class NonLocalBlock(nn.Module):
    """u_net_2d_discriminator.py:335-381."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.theta = EqualizedConv2d(in_channels, out_channels // 8, 1, 1, 0, bias=False)
        self.phi = EqualizedConv2d(in_channels, out_channels // 8, 1, 1, 0, bias=False)
        self.g = EqualizedConv2d(in_channels, out_channels // 2, 1, 1, 0, bias=False)
        self.o = EqualizedConv2d(out_channels // 2, out_channels, 1, 1, 0, bias=False)
        self.residual_mapping = EqualizedConv2d(in_channels, out_channels, 1, 1, 0, bias=False) \
            if in_channels != out_channels else nn.Identity()
        self.gamma = nn.Parameter(torch.tensor(0.))

    def forward(self, x):
        b, _, h, w = x.shape
        theta = self.theta(x).flatten(2)
        phi = F.max_pool2d(self.phi(x), 2, 2).flatten(2)
        g = F.max_pool2d(self.g(x), 2, 2).flatten(2)
        o = self.o(ops.non_local_attention(theta, phi, g).view(b, -1, h, w))
        return (self.gamma * o + self.residual_mapping(x)) / math.sqrt(2)


class Discriminator(nn.Module):
    """u_net_2d_discriminator.py:14-140.  The optional fft input (:43-46, :106-122; off in config.py:12) calls
    `torch.rfft`, which no torch of the last years has: it is restated through torch.fft per channel as the reference
    does it, and that part of the oracle is PARITY UNPINNED (no fixture of the reference can be generated for it)."""

    def __init__(self, config=DISCRIMINATOR_CONFIG, no_rfp=False, no_gfp=False):
        super().__init__()
        enc, dec = config["encoder_channels"], config["decoder_channels"]
        self.fft = bool(config["fft"])
        in_ch = 3 if no_gfp else (6 if no_rfp else 9)
        if self.fft:
            in_ch = in_ch + in_ch * 2
        self.encoder_blocks = nn.ModuleList()
        for i, (a, b) in enumerate(enc):
            if i == 0:
                self.encoder_blocks.append(ResNetBlock(in_ch, b))
            elif i == 2:
                self.encoder_blocks.append(NonLocalBlock(a, b))
            else:
                self.encoder_blocks.append(ResNetBlock(a, b, mini_batch_std_dev=i >= len(enc) - 2))
        self.downscale_convolutions = nn.ModuleList(
            [nn.Sequential(EqualizedConv2d(b, b, 3, 2, 0), Blur()) for _, b in enc[:-1]])
        self.classification_head = nn.Sequential(
            nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(1), EqualizedLinear(enc[-1][-1], 128, bias=False),
            FusedLeakyReLU(128), EqualizedLinear(128, 1, bias=False))
        self.decoder_blocks = nn.ModuleList(
            [NonLocalBlock(a, b) if i == 1 else ResNetBlock(a, b) for i, (a, b) in enumerate(dec)])
        self.transposed_convolutions = nn.ModuleList()
        for cur, past, d in zip(reversed(enc[1:]), reversed(enc[:-1]), dec):
            self.transposed_convolutions.append(nn.Sequential(
                Upsample(), EqualizedConv2d(cur[-1], d[0] - past[-1], 1, 1, 0, bias=False)))
        self.final_mapping = nn.Sequential(FusedLeakyReLU(dec[-1][-1]),
                                           EqualizedConv2d(dec[-1][-1], 1, 1, 1, 0, bias=False))

    def forward(self, x, **kwargs):
        if self.fft:
            # old torch.rfft(v, signal_ndim=3, normalized=True, onesided=False): full complex 3-D DFT / sqrt(T H W),
            # trailing dimension (real, imaginary); the reference moves that dimension in front of T
            spectra = [torch.view_as_real(torch.fft.fftn(x[:, c], dim=(1, 2, 3), norm="ortho")).permute(0, 4, 1, 2, 3)
                       for c in range(x.shape[1])]
            x = torch.cat([x] + spectra, dim=1)
        x = x.flatten(1, 2)
        feats = []
        for i, blk in enumerate(self.encoder_blocks):
            x = blk(x)
            if i != len(self.encoder_blocks) - 1:
                feats.append(x)
                x = self.downscale_convolutions[i](x)
        scalar = self.classification_head(x)
        for blk, up, f in zip(self.decoder_blocks, self.transposed_convolutions, reversed(feats)):
            x = blk(torch.cat([up(x), f], dim=1))
        return scalar, self.final_mapping(x).unsqueeze(2)
