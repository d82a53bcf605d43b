"""U-Net discriminator on the gfx950 kernels, with the reference's nn.Module surface
(multi_stylegan/u_net_2d_discriminator.py:14-381): same class names, constructor / forward signatures and
state_dict keys.  Feature maps run channels-last (optionally bf16); the 6-channel input image and the two
heads' outputs stay fp32."""
import math
import random
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import conv_ops, equalized_layer
from .op_static import FusedLeakyReLU, gamma_merge, max_pool2x2, non_local_attention, scaled_add, scaled_add_fork, softmax_rows, upfirdn2d
from .op_static import attention as _attention
from .op_static import pointwise_head


# Two formulations that tests compare in BOTH positions (bit-identical results; tests/test_hip_models.py); not configuration:
IN_PLACE_CAT = True        # the decoder's concatenations are written in place (False: torch.cat-style copies)
FUSE_INPUT_FORK = True     # a block input's two gradients meet in the 3x3 conv's data-gradient epilogue (False: autograd's add)


def _fir2d(taps, gain=1.0):
    t = torch.tensor(list(taps), dtype=torch.float32)
    k = torch.outer(t, t)
    return k / k.sum() * gain


class Upsample(nn.Module):
    def __init__(self, blur_kernel: List[int] = [1, 3, 3, 1], factor: int = 2) -> None:
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", _fir2d(blur_kernel))
        p = len(blur_kernel) - factor
        self.padding = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """out (not in the reference): the channel-slice of a concatenation buffer that receives the result."""
        return upfirdn2d(input, self.kernel, up=self.factor, pad=self.padding, out=out)


class Blur(nn.Module):
    def __init__(self, kernel: List[int] = [1, 3, 3, 1], sampling_factor: int = 1,
                 sampling_factor_padding: int = 2, kernel_size: int = 3) -> None:
        super().__init__()
        p = (len(kernel) - sampling_factor_padding) + (kernel_size - 1)
        self.padding = ((p + 1) // 2, p // 2)
        self.register_buffer("kernel", _fir2d(kernel, float(sampling_factor ** 2) if sampling_factor > 1 else 1.0))

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return upfirdn2d(input, self.kernel, pad=self.padding)


def _mbstd_composite(input: torch.Tensor, groups: int, alpha: float) -> torch.Tensor:
    """The statistic with torch ops (any device / layout, differentiable to any order); statistics in fp32."""
    b, _, h, w = input.shape
    x = input.float().reshape(groups, b // groups, *input.shape[1:])
    var = (x - x.mean(dim=1, keepdim=True)).square().mean(dim=1)                    # [G, C, H, W]
    stat = torch.sqrt(var.clamp(min=alpha)).mean(dim=(1, 2, 3))                     # [G]
    plane = stat.to(input.dtype).reshape(groups, 1, 1, 1, 1).expand(groups, b // groups, 1, h, w).reshape(b, 1, h, w)
    return conv_ops.cat_channels([input, plane])


class _MinibatchStdDevFused(torch.autograd.Function):
    """cat([x, stat plane]) on the gfx950 kernels (csrc/mbstd.hip): x is read once, the concatenation copy is part of the
    same pass.  A second-order graph (R1) is built from the composite formulation."""

    @staticmethod
    def forward(ctx, x, groups, alpha):
        from . import _lib
        dev = _lib.require_gpu(x)
        b, c, h, w = x.shape
        xv, ldx = conv_ops._nhwc_view(x)
        out = conv_ops._padded_nhwc(b, c + 1, h, w, x.dtype, dev)
        ldy = out.stride(3)
        code_dtype = _lib.dtype_code(x)
        stat = torch.empty(groups, dtype=torch.float32, device=dev)
        work = torch.empty(int(_lib.lib().msg_minibatch_stddev_workspace(c, h, w, groups, code_dtype)),
                           dtype=torch.float32, device=dev)
        with _lib.on_device(dev), _lib.kernel_clock.span(('mbstd_fwd', x.dtype), 2 * x.numel() * x.element_size()):
            code = _lib.lib().msg_minibatch_stddev(xv.data_ptr(), out.data_ptr(), stat.data_ptr(), work.data_ptr(),
                                                   code_dtype, b, c, h, w, ldx, ldy, groups, float(alpha),
                                                   _lib.stream_of(dev))
        _lib.check(code, "msg_minibatch_stddev")
        ctx.save_for_backward(x)
        ctx.cfg = (groups, alpha)
        return out

    @staticmethod
    def backward(ctx, gy):
        from . import _lib
        x, = ctx.saved_tensors
        groups, alpha = ctx.cfg
        if torch.is_grad_enabled():
            with torch.enable_grad():
                xx = x if x.requires_grad else x.detach().requires_grad_(True)
                gx, = torch.autograd.grad(_mbstd_composite(xx, groups, alpha), xx, gy, create_graph=True)
            return gx, None, None
        dev = x.device
        b, c, h, w = x.shape
        xv, ldx = conv_ops._nhwc_view(x)
        gv, ldg = conv_ops._nhwc_view(gy)
        gstat = gy[:, c].float().reshape(groups, -1).sum(dim=1).contiguous()
        gx = torch.empty((b, h, w, c), dtype=x.dtype, device=dev).permute(0, 3, 1, 2)
        with _lib.on_device(dev), _lib.kernel_clock.span(('mbstd_bwd', x.dtype), 3 * x.numel() * x.element_size()):
            code = _lib.lib().msg_minibatch_stddev_backward(xv.data_ptr(), gv.data_ptr(), gstat.data_ptr(), gx.data_ptr(),
                                                            _lib.dtype_code(x), b, c, h, w, ldx, ldg, c, groups,
                                                            float(alpha), _lib.stream_of(dev))
        _lib.check(code, "msg_minibatch_stddev_backward")
        return gx, None, None


class MinibatchStdDev(nn.Module):
    """Appends one plane holding the mean (over c,h,w) of the per-position std over the batch; statistics in
    fp32.  The whole batch of ONE forward call is one group, as in the reference (:205-217); when the trainer runs the
    real and the fake batch through the discriminator as one concatenated batch (Discriminator.forward(...,
    minibatch_groups=2)) each of them keeps its own statistic."""

    def __init__(self, alpha: float = 1e-8) -> None:
        super().__init__()
        self.alpha = alpha
        self.groups = 1        # independent forward batches concatenated along dim 0 (set by Discriminator.forward)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        groups = self.groups
        assert input.shape[0] % groups == 0
        vec = 16 // input.element_size()
        if input.is_cuda and input.dtype in (torch.float32, torch.bfloat16) and input.shape[1] % vec == 0:
            return _MinibatchStdDevFused.apply(input, groups, self.alpha)
        return _mbstd_composite(input, groups, self.alpha)


class ResNetBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, mini_batch_std_dev: bool = False) -> None:
        super().__init__()
        self.mini_batch_std_dev = MinibatchStdDev() if mini_batch_std_dev else nn.Identity()
        self.main_mapping = nn.Sequential(
            equalized_layer.EqualizedConv2d(in_channels + 1 if mini_batch_std_dev else in_channels, out_channels,
                                            kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), bias=False),
            FusedLeakyReLU(out_channels),
            equalized_layer.EqualizedConv2d(out_channels, out_channels, kernel_size=(3, 3), stride=(1, 1),
                                            padding=(1, 1), bias=False),
            FusedLeakyReLU(out_channels))
        self.residual_mapping = equalized_layer.EqualizedConv2d(
            in_channels, out_channels, kernel_size=1, stride=1, padding=0,
            bias=False) if in_channels != out_channels else nn.Identity()

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return self._merge(input, scaled_add)

    def forward_forked(self, input: torch.Tensor, out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """forward() for an output with two consumers (the skip connection and the downscale path): two aliases of the
        result, so that their two gradients are merged and rescaled in one pass (op_static.scaled_add_fork).
        out: where the result may be written (conv_ops.cat_destination: the skip's slice of the decoder's concatenated
        map); used when the merge runs in the residual conv's epilogue -- the caller checks the returned tensor."""
        return self._merge(input, scaled_add_fork, out=out)

    def _merge(self, input: torch.Tensor, merge, out: Optional[torch.Tensor] = None):
        conv1, act1, conv2, act2 = self.main_mapping            # conv -> bias + leaky ReLU fused per pair
        res = self.residual_mapping
        fuse_res = isinstance(res, equalized_layer.EqualizedConv2d) and res.bias is None and input.is_cuda
        slot, x_main, x_res = None, self.mini_batch_std_dev(input), input
        if fuse_res and FUSE_INPUT_FORK and input.requires_grad and x_main is input \
                and conv1.bias is None:
            # the input's two gradients (main 3x3 conv, 1x1 residual conv) meet in the 3x3 conv's data-gradient epilogue
            slot = conv_ops.GradSlot()
            x_main, x_res = conv_ops.fork_input(input, slot)
        # the merge's 1 / sqrt(2) on the main branch's gradient rides in act2's backward when the merge is the fused one
        # and the block output has a single consumer (conv_ops.GradScale)
        owed = conv_ops.GradScale() if fuse_res and \
            merge is not scaled_add_fork and conv2.bias is None and input.is_cuda else None
        # act1's output feeds conv2 and nothing else: its backward rides in conv2's data-gradient epilogue (conv_ops.ActHandle)
        handle = conv_ops.ActHandle() if input.is_cuda and torch.is_grad_enabled() else None
        output = conv2.forward_activated(conv1.forward_activated(x_main, act1, grad_slot=slot, act_handle=handle), act2,
                                         out_grad_scale=owed, input_act=handle)
        if fuse_res and output.dtype == input.dtype:
            # (main + conv1x1(input)) / sqrt(2) in the epilogue of the 1x1 conv: no separate merge pass
            return conv_ops.conv2d_add_residual(x_res, res.weight, output, 1.0 / math.sqrt(2), stride=res.stride,
                                                padding=res.padding, wscale=res.scale, fork=merge is scaled_add_fork,
                                                grad_slot=slot, main_grad_scale=owed, out=out)
        return merge(output, res(x_res), 1.0 / math.sqrt(2))


class NonLocalBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int) -> None:
        super().__init__()
        conv1x1 = lambda i, o: equalized_layer.EqualizedConv2d(i, o, kernel_size=(1, 1), padding=(0, 0), bias=False)
        self.theta = conv1x1(in_channels, out_channels // 8)
        self.phi = conv1x1(in_channels, out_channels // 8)
        self.g = conv1x1(in_channels, out_channels // 2)
        self.o = conv1x1(out_channels // 2, out_channels)
        self.residual_mapping = conv1x1(in_channels, out_channels) if in_channels != out_channels else nn.Identity()
        self.register_parameter(name="gamma", param=nn.Parameter(torch.tensor(0.), requires_grad=True))

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        bsz, _, height, width = input.shape
        # channels-last feature maps ARE [B, HW, C] matrices: queries / values are views, and the second product is
        # taken as beta @ v so that (a) its result is already the NHWC map the next conv reads and (b) beta is used
        # un-transposed in both products (its gradient arrives contiguous: no [B, HW, HW/4] transpose copies).
        projections = [self.theta, self.phi, self.g] + \
            ([self.residual_mapping] if isinstance(self.residual_mapping, equalized_layer.EqualizedConv2d) else [])
        residual = None
        if input.is_cuda and all(m.bias is None and m.stride == (1, 1) for m in projections):
            # the three (four) 1x1 projections of the block input as ONE autograd node: their input gradients meet in the
            # data-gradient launches instead of in three separate accumulation passes over the map
            outs = conv_ops.conv2d_shared_input(input, [(m.weight, m.scale) for m in projections])
            t, p_, g_ = outs[:3]
            residual = outs[3] if len(outs) > 3 else None
        else:
            t, p_, g_ = self.theta(input), self.phi(input), self.g(input)
        query = t.flatten(start_dim=2).transpose(1, 2)                                                  # [B, HW, C/8]
        key = max_pool2x2(p_).flatten(start_dim=2)                                                      # [B, C/8, HW/4]
        value = max_pool2x2(g_).flatten(start_dim=2).transpose(1, 2)                                    # [B, HW/4, C/2]
        keys = key.transpose(1, 2)                                                                      # [B, HW/4, C/8]
        if _attention.supported(query, keys, value):
            # fused: the [B, HW, HW/4] attention map never reaches HBM (csrc/attention.hip)
            attended = non_local_attention(query, keys, value)
        else:
            # softmax accumulates in fp32 whatever the storage type: no fp32 copy of the [B, HW, HW/4] map is made
            scores = torch.bmm(query, key)
            beta = softmax_rows(scores) if input.is_cuda else torch.softmax(scores, dim=-1)
            attended = torch.bmm(beta, value)
        attended = attended.view(bsz, height, width, -1).permute(0, 3, 1, 2)
        output = self.o(conv_ops.to_compute_layout(attended))
        if residual is None:
            residual = self.residual_mapping(input)
        return gamma_merge(output, residual, self.gamma, 1.0 / math.sqrt(2))


def append_spectra(input: torch.Tensor) -> torch.Tensor:
    """[B, C, T, H, W] -> [B, C + 2 C, T, H, W]: behind the C input channels, per channel the real and the imaginary part of
    its orthonormal, two-sided 3-D DFT over (T, H, W) -- what the reference's
    ``torch.rfft(input[:, c], signal_ndim=3, normalized=True, onesided=False).permute(0, 4, 1, 2, 3)`` returned
    (u_net_2d_discriminator.py:109-122; `torch.rfft` no longer exists, so this branch cannot be run from the reference's
    code on any torch this package supports: restated with torch.fft, parity unpinned).  A library FFT (hipFFT), not one of
    this package's kernels: the branch is off in the reference's configuration (config.py:12)."""
    spectrum = torch.view_as_real(torch.fft.fftn(input.float(), dim=(-3, -2, -1), norm="ortho"))     # [B, C, T, H, W, 2]
    spectrum = spectrum.permute(0, 1, 5, 2, 3, 4).flatten(start_dim=1, end_dim=2)                     # [B, 2 C, T, H, W]
    return torch.cat([input, spectrum.to(input.dtype)], dim=1)


class Discriminator(nn.Module):
    supports_minibatch_groups = True     # forward(..., minibatch_groups=n): n concatenated batches, per-batch statistics

    def __init__(self, config: Dict[str, Any], no_rfp: bool = False, no_gfp: bool = False) -> None:
        super().__init__()
        encoder_channels: Tuple[Tuple[int, int], ...] = config["encoder_channels"]
        decoder_channels: Tuple[Tuple[int, int], ...] = config["decoder_channels"]
        self.fft: bool = config["fft"]
        input_channels = 3 if no_gfp else (6 if no_rfp else 9)
        if self.fft:
            # the reference's optional spectral input (u_net_2d_discriminator.py:43-46,106-122; off in config.py:12): every
            # channel's 3-D spectrum (real, imaginary) is appended to the input, tripling the first block's input channels
            input_channels = input_channels + 2 * input_channels
        self.encoder_blocks = nn.ModuleList()
        for index, (c_in, c_out) in enumerate(encoder_channels):
            if index == 0:
                self.encoder_blocks.append(ResNetBlock(input_channels, c_out))
            elif index == 2:
                self.encoder_blocks.append(NonLocalBlock(c_in, c_out))
            else:
                self.encoder_blocks.append(ResNetBlock(c_in, c_out,
                                                       mini_batch_std_dev=index >= len(encoder_channels) - 2))
        self.downscale_convolutions = nn.ModuleList([
            nn.Sequential(equalized_layer.EqualizedConv2d(c, c, kernel_size=(3, 3), stride=(2, 2), padding=(0, 0)),
                          Blur()) for _, c in encoder_channels[:-1]])
        self.classification_head = nn.Sequential(
            nn.AdaptiveAvgPool2d(output_size=(1, 1)), nn.Flatten(start_dim=1),
            equalized_layer.EqualizedLinear(encoder_channels[-1][-1], 128, bias=False),
            FusedLeakyReLU(channel=128),
            equalized_layer.EqualizedLinear(128, 1, bias=False))
        self.decoder_blocks = nn.ModuleList([
            NonLocalBlock(c_in, c_out) if index == 1 else ResNetBlock(c_in, c_out)
            for index, (c_in, c_out) in enumerate(decoder_channels)])
        self.transposed_convolutions = nn.ModuleList()
        for current, past, decoder in zip(reversed(encoder_channels[1:]), reversed(encoder_channels[:-1]),
                                          decoder_channels):
            self.transposed_convolutions.append(nn.Sequential(
                Upsample(),
                equalized_layer.EqualizedConv2d(current[-1], decoder[0] - past[-1], kernel_size=(1, 1),
                                                stride=(1, 1), padding=(0, 0), bias=False)))
        self.final_mapping = nn.Sequential(
            FusedLeakyReLU(channel=decoder_channels[-1][-1]),
            equalized_layer.EqualizedConv2d(decoder_channels[-1][-1], 1, kernel_size=(1, 1), stride=(1, 1),
                                            padding=(0, 0), bias=False))
        self.compute_dtype = torch.float32          # MI355X-side knob; default reproduces the reference

    def forward(self, input: torch.Tensor, minibatch_groups: int = 1, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        """minibatch_groups > 1: `input` is that many forward batches concatenated along dim 0 (the real and the fake
        batch of the discriminator step); everything is per-sample except the minibatch statistic, which is taken per
        group -- the result equals separate forward calls, at the launch count and tile efficiency of one."""
        stats = [m for m in self.modules() if isinstance(m, MinibatchStdDev)] if minibatch_groups != 1 else []
        for m in stats:
            m.groups = minibatch_groups
        try:
            return self._forward(input)
        finally:
            for m in stats:
                m.groups = 1

    def _skip_destination(self, index: int, x: torch.Tensor, block: nn.Module):
        """(buffer, [upsampled slice, skip slice]) of the decoder level that consumes encoder block `index`'s output, or
        None when that level does not take the in-place form."""
        if not (IN_PLACE_CAT and x.is_cuda):
            return None
        up = self.transposed_convolutions[len(self.transposed_convolutions) - 1 - index]
        fir, mix = up[0], up[1]
        if not (isinstance(fir, Upsample) and fir.factor == 2 and isinstance(mix, equalized_layer.EqualizedConv2d)
                and mix.bias is None and mix.kernel_size == (1, 1) and mix.stride == (1, 1) and mix.padding == (0, 0)):
            return None
        conv2 = block.main_mapping[2]
        return conv_ops.cat_destination(x.shape[0], (mix.weight.shape[0], conv2.weight.shape[0]), x.shape[2], x.shape[3],
                                        x.dtype, x.device)

    def _forward(self, input: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.fft:
            input = append_spectra(input)
        x = input.flatten(start_dim=1, end_dim=2)
        x = conv_ops.to_compute_layout(x, self.compute_dtype)
        skips, dests = [], []
        last = len(self.encoder_blocks) - 1
        for index, block in enumerate(self.encoder_blocks):
            if index != last and hasattr(block, "forward_forked"):
                # The skip features are written where the decoder wants them: into their channel-slice of the map that
                # the reference builds with torch.cat([upsampled, skip]) (u_net_2d_discriminator.py:128-131) -- the
                # upsampling FIR later fills the other slice, and the concatenation is no copy at all.
                dest = self._skip_destination(index, x, block)
                skip, x = block.forward_forked(x, out=None if dest is None else dest[1][1])
                if dest is not None and skip.data_ptr() != dest[1][1].data_ptr():
                    dest = None                            # (the block took another path: plain concatenation later)
                skips.append(skip)
                dests.append(dest)
                x = self.downscale_convolutions[index](x)
                continue
            x = block(x)
            if index != last:
                skips.append(x)
                dests.append(None)
                x = self.downscale_convolutions[index](x)
        classification = self.classification_head(x.float()) if x.dtype != torch.float32 else \
            self.classification_head(x)
        for block, up, skip, dest in zip(self.decoder_blocks, self.transposed_convolutions, reversed(skips), reversed(dests)):
            # `up` = Upsample (per-channel FIR) -> bias-free 1x1 conv (per-pixel channel mix): the two commute exactly, and
            # mixing the channels BEFORE upsampling runs the conv on a quarter of the pixels and the FIR on the (fewer)
            # output channels.  Same function as the reference's order (u_net_2d_discriminator.py:120-127) up to rounding.
            fir, mix = up[0], up[1]
            commute = isinstance(mix, equalized_layer.EqualizedConv2d) and mix.bias is None and \
                mix.kernel_size == (1, 1) and mix.stride == (1, 1) and mix.padding == (0, 0)
            if commute and IN_PLACE_CAT and x.is_cuda and isinstance(fir, Upsample):
                low = mix(x)
                if dest is None:                           # (skip produced elsewhere: it is copied, the FIR still writes in place)
                    dest = conv_ops.cat_destination(skip.shape[0], (low.shape[1], skip.shape[1]), skip.shape[2],
                                                    skip.shape[3], skip.dtype, skip.device)
                if dest is not None and low.dtype == skip.dtype and \
                        tuple(dest[1][0].shape) == (low.shape[0], low.shape[1], 2 * low.shape[2], 2 * low.shape[3]):
                    x = block(conv_ops.cat_in_place(dest[0], [fir(low, out=dest[1][0]), skip]))
                    continue
                x = block(conv_ops.cat_channels([fir(low), skip]))
                continue
            x = block(conv_ops.cat_channels([fir(mix(x)) if commute else up(x), skip]))
        act, conv = self.final_mapping
        if pointwise_head.supported(x, conv, act):
            # activation + 1x1 conv to the one plane in one pass over the map (and one in backward): op_static.pointwise_head
            pixel_wise = pointwise_head.act_pointwise_head(x, act, conv).unsqueeze(dim=2)
        else:
            pixel_wise = self.final_mapping(x).float().contiguous().unsqueeze(dim=2)
        return classification, pixel_wise


# ---------------------------------------------------------------------------------------------------- CutMix
# (reference u_net_2d_discriminator.py:384-448; used by ModelWrapper's late-training branch, model_wrapper.py:331-376)
def _generate_binary_cut_mix_map(height: int, width: int, device: Union[str, torch.device] = "cpu") -> torch.Tensor:
    """Random two-region binary map [1,1,1,H,W]: one corner rectangle against the rest, randomly inverted.  The random
    draws are the reference's, in its order (two torch.randint on the CPU generator for the cut row / column, then two
    random.random() for the corner and the inversion), so a seeded run reproduces the reference's maps; the map itself
    is built on `device` from two comparisons instead of slice assignments."""
    cut_row = int(torch.randint(int(0.1 * height), int(0.9 * height), size=(1,)))
    cut_col = int(torch.randint(int(0.1 * width), int(0.9 * width), size=(1,)))
    rows = torch.arange(height, device=device).view(height, 1)
    cols = torch.arange(width, device=device).view(1, width)
    lower_right = random.random() > 0.5
    region = (rows >= cut_row) & (cols >= cut_col) if lower_right else (rows < cut_row) & (cols < cut_col)
    binary_map = region.to(torch.float).view(1, 1, 1, height, width)
    if random.random() > 0.5:
        binary_map = 1. - binary_map
    return binary_map


def generate_cut_mix_augmentation_data(image_real: torch.Tensor, image_fake: torch.Tensor,
                                       binary_map: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Mixed image (real where the map is 1, fake elsewhere) and the map as its per-pixel real/fake label.  The fake
    batch may be longer than the real one (wrongly ordered reals appended); it is cut to the real batch's length."""
    image_fake = image_fake[:image_real.shape[0]]
    if binary_map is None:
        binary_map = _generate_binary_cut_mix_map(image_real.shape[-2], image_fake.shape[-1], image_real.device)
    keep = binary_map.to(image_real.dtype)
    return image_real * keep + image_fake * (1. - keep), binary_map


def generate_cut_mix_transformation_data(image_real: torch.Tensor, image_fake: torch.Tensor,
                                         prediction_real: torch.Tensor, prediction_fake: torch.Tensor,
                                         binary_map: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Mixed image and the equally mixed pixel-wise predictions (the soft target of the consistency term)."""
    count = image_real.shape[0]
    image_fake, prediction_fake = image_fake[:count], prediction_fake[:count]
    if binary_map is None:
        binary_map = _generate_binary_cut_mix_map(image_real.shape[-2], image_fake.shape[-1], image_real.device)
    keep = binary_map.to(image_real.dtype)
    return image_real * keep + image_fake * (1. - keep), prediction_real * keep + prediction_fake * (1. - keep)
