#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python modules on CPU.

Runs only in the build container (needs /root/reference); nothing of the
reference travels with the fixtures -- they are plain input/output tensors.

How the reference is made importable: its two CUDA
extension modules are not built here, and the package ``__init__`` pulls in
torchvision/kornia/rtpt which are not installed.  So
  * ``multi_stylegan`` is registered as a bare namespace package pointing at the
    reference directory (skips ``__init__``),
  * ``upfirdn2d_cuda.upfirdn2d`` forwards to the reference's OWN
    ``upfirdn2d_native`` (same 10-argument signature, op_static/upfirdn2d.py:156),
  * ``fused_act_cuda.fused_bias_act`` is a three-line torch statement of
    fused_bias_act_kernel.cu:26-47 (the only arithmetic of the path that cannot be
    executed from the reference's own code here),
  * ``torchvision`` (imported by misc.py for image dumps only) is an empty module.
While generating, every vector is also compared with ``oracle/`` and the script
fails if the restatement disagrees.

Usage: python tools/gen_golden.py [--check-only]
"""
import argparse
import importlib
import json
import math
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def import_reference():
    sys.path.insert(0, REF)
    pkg = types.ModuleType("multi_stylegan")
    pkg.__path__ = [os.path.join(REF, "multi_stylegan")]
    sys.modules["multi_stylegan"] = pkg
    up_mod, act_mod = types.ModuleType("upfirdn2d_cuda"), types.ModuleType("fused_act_cuda")
    sys.modules["upfirdn2d_cuda"], sys.modules["fused_act_cuda"] = up_mod, act_mod

    def fused_bias_act(x, b, ref, act, grad, alpha, scale):
        assert act == 3
        if b.numel():
            x = x + b.view(1, -1, *([1] * (x.ndim - 2)))
        gate = ref if grad == 1 else x
        return torch.where(gate > 0, x, x * alpha) * scale

    act_mod.fused_bias_act = fused_bias_act
    up_mod.upfirdn2d = lambda *a: sys.modules["multi_stylegan.op_static.upfirdn2d"].upfirdn2d_native(*a)
    # misc.py imports torchvision (absent here) for Logger.save_prediction only; an empty module lets the hot-path
    # helpers random_permutation / get_noise / exponential_moving_average be taken from the reference itself
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    names = ["op_static.upfirdn2d", "op_static.fused_act", "equalized_layer", "multi_stylegan_generator",
             "u_net_2d_discriminator", "loss", "config", "misc"]
    return {n.split(".")[-1]: importlib.import_module("multi_stylegan." + n) for n in names}


def npy(t):
    return t.detach().cpu().numpy().copy()      # copy: params are updated in place later


def close(a, b, tol=2e-5, what=""):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    ref = b.abs().max().item()
    err = (a - b).abs().max().item() / ref if ref > 0 else (a - b).abs().max().item()      # no floor (as tests/conftest.rel_err)
    assert err <= tol, f"oracle != reference for {what}: rel err {err:.3e}"
    return err


TINY_G = {"channels": (16, 16, 16, 16), "channel_factor": 1, "latent_dimensions": 16,
          "depth_style_mapping": 2, "starting_resolution": (4, 4)}
TINY_D = {"encoder_channels": ((3, 8), (8, 16), (16, 24), (24, 32), (32, 48)),
          "decoder_channels": ((48, 32), (32, 24), (24, 16), (16, 8)), "fft": False}


def gen_upfirdn(ref, oracle_ops, store):
    up = ref["upfirdn2d"]
    g = torch.Generator().manual_seed(11)
    blur4 = oracle_ops.make_fir((1, 3, 3, 1), 4.0)
    blur1 = oracle_ops.make_fir((1, 3, 3, 1), 1.0)
    asym = torch.randn(4, 4, generator=g)
    cases = {  # name: (shape, fir, up, down, pad)
        "g_blur_pad21_gain4": ((2, 3, 8, 8), blur4, 1, 1, (2, 1)),
        "g_skip_up2_pad21": ((2, 3, 5, 7), blur1, 2, 1, (2, 1)),
        "d_blur_pad22_odd": ((1, 4, 15, 15), blur1, 1, 1, (2, 2)),
        "bwd_of_up2_down2": ((2, 2, 10, 14), blur1, 1, 2, (1, 1)),
        "asym_blur_pad21": ((1, 2, 6, 9), asym, 1, 1, (2, 1)),
        "asym_up2_pad21": ((1, 2, 4, 6), asym, 2, 1, (2, 1)),
        "asym_down2_pad12": ((1, 2, 9, 12), asym, 1, 2, (1, 2)),
        "blur_pad11": ((1, 1, 7, 7), blur1, 1, 1, (1, 1)),
    }
    for name, (shape, fir, u, d, pad) in cases.items():
        x = torch.randn(*shape, generator=g, requires_grad=True)
        y = up.upfirdn2d(x, fir, up=u, down=d, pad=pad)
        gy = torch.randn(y.shape, generator=g, requires_grad=True)
        gx, = torch.autograd.grad(y, x, gy, create_graph=True)
        ggx = torch.randn(x.shape, generator=g)
        ggy, = torch.autograd.grad(gx, gy, ggx)
        xo = x.detach().clone().requires_grad_(True)
        yo = oracle_ops.upfirdn2d(xo, fir, up=u, down=d, pad=pad)
        gyo = gy.detach().clone().requires_grad_(True)
        gxo, = torch.autograd.grad(yo, xo, gyo, create_graph=True)
        ggyo, = torch.autograd.grad(gxo, gyo, ggx)
        close(yo, y, what=name + ".y"); close(gxo, gx, what=name + ".gx"); close(ggyo, ggy, what=name + ".ggy")
        # literal CUDA index math on the same input
        lit = oracle_ops.upfirdn2d_scalar(x.detach().reshape(-1, *shape[2:]), fir, u, d, pad[0], pad[1])
        close(lit.reshape(y.shape), y, what=name + ".scalar")
        store.update({f"{name}.x": npy(x), f"{name}.fir": npy(fir), f"{name}.cfg": np.array([u, d, *pad]),
                      f"{name}.y": npy(y), f"{name}.gy": npy(gy), f"{name}.gx": npy(gx),
                      f"{name}.ggx": npy(ggx), f"{name}.ggy": npy(ggy)})


def gen_fused_act(ref, oracle_ops, store):
    fa = ref["fused_act"]
    g = torch.Generator().manual_seed(12)
    for name, shape, scale in (("mlp_2d", (4, 16), 1.0), ("conv_4d", (2, 6, 5, 7), 1.0),
                               ("conv_4d_sqrt2", (2, 3, 4, 4), 2 ** 0.5)):
        x = torch.randn(*shape, generator=g, requires_grad=True)
        b = torch.randn(shape[1], generator=g, requires_grad=True)
        y = fa.fused_leaky_relu(x, b, 0.2, scale)
        gy = torch.randn(y.shape, generator=g, requires_grad=True)
        gx, gb = torch.autograd.grad(y, (x, b), gy, create_graph=True)
        ggx, ggb = torch.randn(x.shape, generator=g), torch.randn(b.shape, generator=g)
        ggy, = torch.autograd.grad((gx, gb), gy, (ggx, ggb))
        xo, bo = x.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
        yo = oracle_ops.fused_leaky_relu(xo, bo, 0.2, scale)
        gyo = gy.detach().clone().requires_grad_(True)
        gxo, gbo = torch.autograd.grad(yo, (xo, bo), gyo, create_graph=True)
        ggyo, = torch.autograd.grad((gxo, gbo), gyo, (ggx, ggb))
        for a, r, w in ((yo, y, "y"), (gxo, gx, "gx"), (gbo, gb, "gb"), (ggyo, ggy, "ggy")):
            close(a, r, what=f"fused_act.{name}.{w}")
        store.update({f"{name}.x": npy(x), f"{name}.b": npy(b), f"{name}.scale": np.array(scale),
                      f"{name}.y": npy(y), f"{name}.gy": npy(gy), f"{name}.gx": npy(gx), f"{name}.gb": npy(gb),
                      f"{name}.ggx": npy(ggx), f"{name}.ggb": npy(ggb), f"{name}.ggy": npy(ggy)})


def load_matching(dst, src):
    missing = dst.load_state_dict(src.state_dict(), strict=True)
    return missing


def gen_modconv(ref, om, store):
    G = ref["multi_stylegan_generator"]
    g = torch.Generator().manual_seed(13)
    kinds = {"conv3x3_demod": dict(kernel_size=(3, 3), demodulate=True, upsampling=False),
             "up2x2_demod": dict(kernel_size=(2, 2), demodulate=True, upsampling=True),
             "torgb1x1_nodemod": dict(kernel_size=(1, 1), demodulate=False, upsampling=False)}
    for kname, kw in kinds.items():
        for mm in (True, False):
            name = f"{kname}.{'map' if mm else 'nomap'}"
            out_c = 3 if kname.startswith("torgb") else 12
            m = G.ModulatedConv2d(in_channels=8, out_channels=out_c, style_dimension=10,
                                  modulation_mapping=mm, **kw)
            with torch.no_grad():
                for p in m.parameters():
                    p.copy_(torch.randn(p.shape, generator=g) * (1.0 if p.ndim > 1 else 0.5) + (p.ndim == 1))
            mo = om.ModulatedConv2d(8, out_c, 10, modulation_mapping=mm, **kw)
            load_matching(mo, m)
            x = torch.randn(2, 8, 6, 6, generator=g, requires_grad=True)
            st = torch.randn(2, 10, generator=g, requires_grad=True) if mm else \
                (torch.randn(2, 1, 8, 1, 1, generator=g) + 1.0).requires_grad_(True)
            res = m(x, st)
            y = res[0] if mm else res
            gy = torch.randn(y.shape, generator=g)
            params = [m.weight] + ([m.modulation_mapping.weight, m.modulation_mapping.bias] if mm else [])
            grads = torch.autograd.grad(y, [x, st] + params, gy, create_graph=True)
            # one second-order quantity: d/d(style) of |grad_x|^2 (what path-length / R1 style terms need)
            gg_st, = torch.autograd.grad(grads[0].square().sum(), st, retain_graph=True)
            xo, sto = x.detach().clone().requires_grad_(True), st.detach().clone().requires_grad_(True)
            reso = mo(xo, sto)
            yo = reso[0] if mm else reso
            po = [mo.weight] + ([mo.modulation_mapping.weight, mo.modulation_mapping.bias] if mm else [])
            gro = torch.autograd.grad(yo, [xo, sto] + po, gy, create_graph=True)
            gg_sto, = torch.autograd.grad(gro[0].square().sum(), sto, retain_graph=True)
            close(yo, y, what=name + ".y")
            for a, r in zip(gro, grads):
                close(a, r, what=name + ".grad")
            close(gg_sto, gg_st, tol=1e-4, what=name + ".gg_style")
            if mm:
                close(reso[1], res[1], what=name + ".style_out")
                store[f"{name}.style_out"] = npy(res[1])
            store.update({f"{name}.x": npy(x), f"{name}.style": npy(st), f"{name}.y": npy(y), f"{name}.gy": npy(gy),
                          f"{name}.gx": npy(grads[0]), f"{name}.gstyle": npy(grads[1]),
                          f"{name}.gweight": npy(grads[2]), f"{name}.gg_style": npy(gg_st)})
            for k, v in m.state_dict().items():
                store[f"{name}.sd.{k}"] = npy(v)


def gen_layers(ref, om, oracle_ops, store):
    E, D = ref["equalized_layer"], ref["u_net_2d_discriminator"]
    g = torch.Generator().manual_seed(14)
    lin = E.EqualizedLinear(12, 7, bias=True)
    with torch.no_grad():
        lin.bias.copy_(torch.randn(7, generator=g))
    x = torch.randn(3, 12, generator=g)
    y = lin(x)
    close(oracle_ops.equalized_linear(x, lin.weight, lin.bias), y, what="eqlinear")
    store.update({"eqlinear.x": npy(x), "eqlinear.w": npy(lin.weight), "eqlinear.b": npy(lin.bias), "eqlinear.y": npy(y)})
    for name, kw in (("eqconv3x3", dict(kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), bias=False)),
                     ("eqconv3x3_s2_bias", dict(kernel_size=(3, 3), stride=(2, 2), padding=(0, 0), bias=True)),
                     ("eqconv1x1", dict(kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), bias=False))):
        conv = E.EqualizedConv2d(6, 10, **kw)
        if conv.bias is not None:
            with torch.no_grad():
                conv.bias.copy_(torch.randn(10, generator=g))
        x = torch.randn(2, 6, 9, 9, generator=g, requires_grad=True)
        y = conv(x)
        gy = torch.randn(y.shape, generator=g)
        gx, gw = torch.autograd.grad(y, (x, conv.weight), gy)
        yo = oracle_ops.equalized_conv2d(x, conv.weight, conv.bias, kw["stride"][0], kw["padding"][0])
        close(yo, y, what=name)
        store.update({f"{name}.x": npy(x), f"{name}.w": npy(conv.weight), f"{name}.y": npy(y), f"{name}.gy": npy(gy),
                      f"{name}.gx": npy(gx), f"{name}.gw": npy(gw),
                      f"{name}.cfg": np.array([kw["stride"][0], kw["padding"][0]])})
        if conv.bias is not None:
            store[f"{name}.b"] = npy(conv.bias)
    for name, mod, x, fn in (
            ("eqconvT2x2", E.EqualizedTransposedConv2d(6, 10, kernel_size=2, stride=2, padding=0, bias=True),
             torch.randn(2, 6, 5, 7, generator=g), oracle_ops.equalized_conv_transpose2d),
            ("eqconv1d", E.EqualizedConv1d(6, 10, kernel_size=3, stride=1, padding=1, bias=True),
             torch.randn(2, 6, 11, generator=g), oracle_ops.equalized_conv1d),
            ("eqconv1d_s2", E.EqualizedConv1d(6, 10, kernel_size=5, stride=2, padding=2, bias=True),
             torch.randn(2, 6, 12, generator=g), lambda a, w, b: oracle_ops.equalized_conv1d(a, w, b, stride=2, padding=2))):
        with torch.no_grad():
            mod.bias.copy_(torch.randn(mod.bias.shape, generator=g))
        x.requires_grad_(True)
        y = mod(x)
        gy = torch.randn(y.shape, generator=g)
        gx, gw = torch.autograd.grad(y, (x, mod.weight), gy)
        close(fn(x, mod.weight, mod.bias), y, what=name)
        store.update({f"{name}.x": npy(x), f"{name}.w": npy(mod.weight), f"{name}.b": npy(mod.bias), f"{name}.y": npy(y),
                      f"{name}.gy": npy(gy), f"{name}.gx": npy(gx), f"{name}.gw": npy(gw)})
    x = torch.randn(3, 16, generator=g)
    y = E.PixelwiseNormalization()(x)
    close(oracle_ops.pixel_norm(x), y, what="pixelnorm")
    store.update({"pixelnorm.x": npy(x), "pixelnorm.y": npy(y)})
    x = torch.randn(4, 5, 3, 3, generator=g)
    y = D.MinibatchStdDev()(x)
    close(oracle_ops.minibatch_stddev(x), y, what="mbstd")
    store.update({"mbstd.x": npy(x), "mbstd.y": npy(y)})
    for name, ctor, octor in (("nonlocal", lambda: D.NonLocalBlock(8, 16), lambda: om.NonLocalBlock(8, 16)),
                              ("resnet_mbstd", lambda: D.ResNetBlock(8, 12, mini_batch_std_dev=True),
                               lambda: om.ResNetBlock(8, 12, mini_batch_std_dev=True))):
        blk, oblk = ctor(), octor()
        with torch.no_grad():
            for p in blk.parameters():
                if p.ndim <= 1:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.7)
        load_matching(oblk, blk)
        x = torch.randn(2, 8, 6, 6, generator=g, requires_grad=True)
        y = blk(x)
        gy = torch.randn(y.shape, generator=g)
        gx, = torch.autograd.grad(y, x, gy)
        close(oblk(x), y, what=name)
        store.update({f"{name}.x": npy(x), f"{name}.y": npy(y), f"{name}.gy": npy(gy), f"{name}.gx": npy(gx)})
        for k, v in blk.state_dict().items():
            store[f"{name}.sd.{k}"] = npy(v)


def perturb_small_params(model, g):
    """Move zero-initialised scalars/biases off their init so they are actually exercised."""
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("noise_injection.weight") or n.endswith("gamma"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
            elif n.endswith("activation.bias") or (n.endswith(".bias") and p.ndim == 1) or p.shape == (1, 1, 1, 1):
                p.add_(torch.randn(p.shape, generator=g) * 0.3)
            elif n.endswith(".input"):
                p.add_(torch.randn(p.shape, generator=g) * 0.5)


def fixed_noise(g, levels, bsz=1):
    out = [torch.randn(bsz, 1, 4, 4, generator=g)]
    for i in range(levels):
        r = 2 ** (i + 3)
        out += [torch.randn(bsz, 1, r, r, generator=g), torch.randn(bsz, 1, r, r, generator=g)]
    return out


def gen_tiny_models(ref, om, store):
    G, D = ref["multi_stylegan_generator"], ref["u_net_2d_discriminator"]
    g = torch.Generator().manual_seed(15)
    torch.manual_seed(15)
    gen, dis = G.Generator(TINY_G), D.Discriminator(TINY_D, no_rfp=True)
    perturb_small_params(gen, g); perturb_small_params(dis, g)
    ogen, odis = om.Generator(TINY_G), om.Discriminator(TINY_D, no_rfp=True)
    load_matching(ogen, gen); load_matching(odis, dis)
    z = [torch.randn(3, 16, generator=g), torch.randn(3, 16, generator=g)]
    noise = fixed_noise(g, 3, bsz=3)
    img, lat = gen(z, return_main_style_vectors=True, noise=noise, inject_index=3)
    oimg, olat = ogen(z, return_main_style_vectors=True, noise=noise, inject_index=3)
    close(oimg, img, what="tinyG.image"); close(olat, lat, what="tinyG.latent")
    gimg = torch.randn(img.shape, generator=g)
    gen.zero_grad(); img.backward(gimg)
    ogen.zero_grad(); oimg.backward(gimg)
    none_grad = sorted(n for n, p in gen.named_parameters() if p.grad is None)
    assert none_grad == sorted(n for n, p in ogen.named_parameters() if p.grad is None)
    assert all(n.startswith("main_convolutions_2.") for n in none_grad) and none_grad
    pick_g = ["style_mapping.layers.1.weight", "main_convolutions_1.5.modulated_convolution.weight",
              "main_convolutions_1.0.modulated_convolution.modulation_mapping.weight",
              "starting_convolution_2.modulated_convolution.weight", "output_blocks_2.2.modulated_convolution.weight",
              "main_convolutions_1.4.noise_injection.weight", "main_convolutions_1.3.activation.bias",
              "constant_input_1.input", "output_blocks_1.1.bias"]
    gp, ogp = dict(gen.named_parameters()), dict(ogen.named_parameters())
    for n in pick_g:
        close(ogp[n].grad, gp[n].grad, tol=1e-4, what="tinyG.grad." + n)
        store["tinyG.grad." + n] = npy(gp[n].grad)
    # path-length style double backward: d/dtheta || d(image.noise)/dlatent ||
    pn = torch.randn(img.shape, generator=g)

    def pl_term(model):
        im, la = model(z, return_main_style_vectors=True, noise=noise, inject_index=3)
        gr, = torch.autograd.grad((im * pn).sum() / math.sqrt(3 * 32 * 32), la, create_graph=True)
        return gr, torch.sqrt(gr.pow(2).sum(2).mean(1) + 1e-8).mean()
    gen.zero_grad(); ogen.zero_grad()
    gr, pl = pl_term(gen); pl.backward()
    ogr, opl = pl_term(ogen); opl.backward()
    close(ogr, gr, tol=1e-4, what="tinyG.pl_grads"); close(opl, pl, tol=1e-4, what="tinyG.pl")
    for n in pick_g[:5]:
        close(ogp[n].grad, gp[n].grad, tol=2e-4, what="tinyG.plgrad." + n)
        store["tinyG.plgrad." + n] = npy(gp[n].grad)
    store.update({"tinyG.z0": npy(z[0]), "tinyG.z1": npy(z[1]), "tinyG.image": npy(img), "tinyG.latent": npy(lat),
                  "tinyG.gimage": npy(gimg), "tinyG.pl_image_noise": npy(pn), "tinyG.pl_grads": npy(gr),
                  "tinyG.pl": npy(pl)})
    for i, t in enumerate(noise):
        store[f"tinyG.noise{i}"] = npy(t)
    for k, v in gen.state_dict().items():
        store["tinyG.sd." + k] = npy(v)
    store_json = {"tinyG.none_grad": none_grad}
    # discriminator
    xin = torch.rand(3, 2, 3, 32, 32, generator=g).requires_grad_(True)
    s, px = dis(xin)
    os_, opx = odis(xin)
    close(os_, s, what="tinyD.scalar"); close(opx, px, what="tinyD.pixel")
    dis.zero_grad(); odis.zero_grad()
    gs, gpx = torch.randn(s.shape, generator=g), torch.randn(px.shape, generator=g)
    gin, = torch.autograd.grad((s, px), xin, (gs, gpx), create_graph=True)
    ogin, = torch.autograd.grad((os_, opx), xin, (gs, gpx), create_graph=True)
    close(ogin, gin, tol=1e-4, what="tinyD.gin")
    r1 = 0.5 * gin.pow(2).reshape(3, -1).sum(1).mean(); r1.backward()
    or1 = 0.5 * ogin.pow(2).reshape(3, -1).sum(1).mean(); or1.backward()
    pick_d = ["encoder_blocks.0.main_mapping.0.weight", "encoder_blocks.2.theta.weight", "encoder_blocks.2.gamma",
              "downscale_convolutions.1.0.weight", "downscale_convolutions.1.0.bias",
              "decoder_blocks.3.main_mapping.2.weight", "encoder_blocks.4.main_mapping.1.bias",
              "transposed_convolutions.0.1.weight"]
    dp, odp = dict(dis.named_parameters()), dict(odis.named_parameters())
    for n in pick_d:
        close(odp[n].grad, dp[n].grad, tol=2e-4, what="tinyD.r1grad." + n)
        store["tinyD.r1grad." + n] = npy(dp[n].grad)
    store.update({"tinyD.x": npy(xin), "tinyD.scalar": npy(s), "tinyD.pixel": npy(px), "tinyD.gs": npy(gs),
                  "tinyD.gpx": npy(gpx), "tinyD.gin": npy(gin), "tinyD.r1": npy(r1)})
    for k, v in dis.state_dict().items():
        store["tinyD.sd." + k] = npy(v)
    return store_json, (gen, dis)


def gen_train_step(ref, om, ot, store):
    """Two iterations (the 2nd with iteration=16 so both lazy regularisers fire), reference modules driven by the
    step order of model_wrapper.py:253-451 (the wrapper itself needs rtpt/tqdm/torchvision and cannot be imported).

    Besides losses and post-step parameters, every optimiser step is recorded whole: the pre-clip gradient of every
    parameter, the global gradient norm clip_grad_norm_ returns, and the movement p_after - p_before of every
    parameter; the EMA copy starts AWAY from the generator (G0 + 0.05 randn) so that its movement per iteration
    (0.001 (p - ema) ~ 5e-5) is far above fp32 resolution and a missing / doubled EMA step is visible.

    ``zero_grad(set_to_none=False)``: the reference calls ``optimizer.zero_grad()`` under its pinned torch 1.8.1
    (requirements.txt:1), where that ZEROES gradients; torch 2's default would drop them and make Adam skip the
    parameters a regulariser's graph does not reach."""
    G, D, L = ref["multi_stylegan_generator"], ref["u_net_2d_discriminator"], ref["loss"]
    import copy
    g = torch.Generator().manual_seed(16)
    torch.manual_seed(16)
    gen, dis = G.Generator(TINY_G), D.Discriminator(TINY_D, no_rfp=True)
    perturb_small_params(gen, g); perturb_small_params(dis, g)
    ogen, odis = om.Generator(TINY_G), om.Discriminator(TINY_D, no_rfp=True)
    load_matching(ogen, gen); load_matching(odis, dis)
    gen_ema = copy.deepcopy(gen)
    with torch.no_grad():
        for p in gen_ema.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
    ogen_ema = copy.deepcopy(ogen)
    load_matching(ogen_ema, gen_ema)
    for k, v in gen.state_dict().items():
        store["train.G0." + k] = npy(v)
    for k, v in gen_ema.state_dict().items():
        store["train.Gema0." + k] = npy(v)
    for k, v in dis.state_dict().items():
        store["train.D0." + k] = npy(v)
    hp = ref["config"].generation_hyperparameters
    og = torch.optim.Adam(gen.get_parameters(2e-4, 2e-6), betas=hp["betas"])
    od = torch.optim.Adam(dis.parameters(), lr=6e-4, betas=hp["betas"])
    oog, ood = ot.make_optimizers(ogen, odis)
    pl_ref, pl_or = L.PathLengthRegularization(), ot.PathLength()
    d_loss, g_loss, r1_loss = L.NonSaturatingLogisticDiscriminatorLoss(), L.NonSaturatingLogisticGeneratorLoss(), \
        L.R1Regularization()
    bsz = 4

    def clip_step(model, optimizer, trace, label):      # model_wrapper.py:296-298 and its three siblings, recorded
        named = [(n, p) for n, p in model.named_parameters() if p.grad is not None]
        before = {n: p.detach().clone() for n, p in named}
        for n, p in named:
            trace[f"{label}.grad.{n}"] = p.grad.detach().clone()
        trace[f"{label}.gnorm"] = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=5.).detach().clone()
        optimizer.step()
        for n, p in named:
            trace[f"{label}.delta.{n}"] = p.detach() - before[n]

    M, U = ref["misc"], ref["u_net_2d_discriminator"]
    top_k_ref, top_k_or = L.TopK(starting_iteration=0, final_iteration=1), ot.TopK(0, 1)
    cm_loss = L.NonSaturatingLogisticDiscriminatorLossCutMix()
    # third iteration: the late-training branches, as `--resume_training` switches them on (model_wrapper.py:121-123,
    # 272, 331-332): wrongly ordered reals among the fakes, CutMix augmentation + consistency, top-k with v = 0.5
    for step, (iteration, late) in enumerate(((1, False), (16, False), (32, True))):
        real = torch.rand(bsz, 2, 3, 32, 32, generator=g)
        dr = ot.Draws(
            z_d=[torch.randn(bsz, 16, generator=g) for _ in range(2)], inject_d=2, noise_d=fixed_noise(g, 3, bsz),
            z_g=torch.randn(bsz, 16, generator=g), noise_g=fixed_noise(g, 3, bsz),
            z_pl=[torch.randn(bsz // 2, 16, generator=g) for _ in range(2)], inject_pl=4,
            noise_pl=fixed_noise(g, 3, bsz // 2), pl_image_noise=torch.randn(bsz // 2, 2, 3, 32, 32, generator=g))
        pre = f"train.it{step}."
        store[pre + "real"] = npy(real)
        for key in ("z_d", "z_g", "z_pl", "noise_d", "noise_g", "noise_pl"):
            v = getattr(dr, key)
            for i, t in enumerate(v if isinstance(v, list) else [v]):
                store[f"{pre}{key}.{i}"] = npy(t)
        store[pre + "pl_image_noise"] = npy(dr.pl_image_noise)
        # ---- reference sequence
        log, trace = {}, {}
        od.zero_grad(set_to_none=False); og.zero_grad(set_to_none=False)
        with torch.no_grad():
            fake = gen(input=dr.z_d, inject_index=dr.inject_d, noise=dr.noise_d)
        if late:
            np.random.seed(300 + step)
            perm = M.random_permutation(real.shape[2])
            np.random.seed(300 + step)
            assert torch.equal(ot.random_permutation(real.shape[2]), perm)
            dr.wrong_order_perm = perm
            store[pre + "wrong_order_perm"] = npy(perm)
            fake = torch.cat([fake, real[:max(1, int(hp["batch_factor_wrong_order"] * real.shape[0])), :, perm]], dim=0)
        pr, prp = dis(real, is_real=True, is_cut_mix=False)
        pf, pfp = dis(fake, is_real=False, is_cut_mix=False)
        lr_, lf = d_loss(pr, pf); lrp, lfp = d_loss(prp, pfp)
        (lr_ + lf + lrp + lfp).backward()
        clip_step(dis, od, trace, "d")
        log.update(loss_d_real=lr_.item(), loss_d_fake=lf.item(), loss_d_real_px=lrp.item(), loss_d_fake_px=lfp.item())
        if iteration % hp["lazy_discriminator_regularization"] == 0:
            od.zero_grad(set_to_none=False); og.zero_grad(set_to_none=False)
            rr = real.clone().requires_grad_(True)
            pr, prp = dis(rr)                       # overwrites the D step's real predictions, as the reference does
            r1 = r1_loss(pr, rr, prp)
            (hp["w_discriminator_regularization_r1"] * r1).backward()
            clip_step(dis, od, trace, "r1")
            log["r1"] = r1.item()
        if late:                                    # model_wrapper.py:331-376
            import random as pyrandom
            real_cm = rr if iteration % hp["lazy_discriminator_regularization"] == 0 else real

            def seeded(fn, seed):                   # the map generators draw from torch's CPU RNG and Python's
                state = torch.get_rng_state()
                torch.manual_seed(seed); pyrandom.seed(seed)
                out = fn()
                torch.set_rng_state(state)
                return out
            od.zero_grad(set_to_none=False); og.zero_grad(set_to_none=False)
            cm_images, cm_label = seeded(lambda: U.generate_cut_mix_augmentation_data(real_cm, fake), 400 + step)
            map_aug = seeded(lambda: U._generate_binary_cut_mix_map(32, 32), 400 + step)
            assert torch.equal(map_aug, cm_label)
            assert torch.equal(seeded(lambda: ot.binary_cut_mix_map(32, 32), 400 + step), map_aug)
            _, cm_pred = dis(cm_images, is_cut_mix=True)
            cm_r, cm_f = cm_loss(cm_pred, cm_label)
            (hp["w_discriminator_regularization"] * (cm_r + cm_f)).backward()
            clip_step(dis, od, trace, "cm_aug")
            log["cut_mix_aug"] = (cm_r + cm_f).item()
            od.zero_grad(set_to_none=False)
            cr_images, cr_label = seeded(lambda: U.generate_cut_mix_transformation_data(
                real_cm.detach(), fake.detach(), prp.detach(), pfp.detach()), 500 + step)
            map_reg = seeded(lambda: U._generate_binary_cut_mix_map(32, 32), 500 + step)
            assert torch.equal(cr_images, real_cm.detach() * map_reg + fake.detach()[:bsz] * (1. - map_reg))
            _, cr_pred = dis(cr_images, is_cut_mix=True)
            cr = torch.nn.functional.mse_loss(cr_pred, cr_label, reduction="mean")
            (hp["w_discriminator_regularization"] * cr).backward()
            clip_step(dis, od, trace, "cm_reg")
            log["cut_mix_reg"] = cr.item()
            dr.cut_mix, dr.cut_mix_map_aug, dr.cut_mix_map_reg = True, map_aug, map_reg
            store[pre + "cut_mix_map_aug"], store[pre + "cut_mix_map_reg"] = npy(map_aug), npy(map_reg)
            store[pre + "cut_mix_seeds"] = np.array([400 + step, 500 + step])
        else:
            dr.cut_mix = False
        od.zero_grad(set_to_none=False); og.zero_grad(set_to_none=False)
        fake = gen(input=dr.z_g, noise=dr.noise_g)
        pf, pfp = dis(fake)
        if late:                                    # model_wrapper.py:392-401 with the resumed TopK(0, 1): v = 0.5
            pf, indexes = top_k_ref(pf)
            pfp = pfp[indexes]
            store[pre + "top_k_indexes"] = npy(indexes)
        lg, lgp = g_loss(pf), g_loss(pfp)
        (lg + lgp).backward()
        clip_step(gen, og, trace, "g")
        log.update(loss_g=lg.item(), loss_g_px=lgp.item())
        if iteration % hp["lazy_generator_regularization"] == 0:
            od.zero_grad(set_to_none=False); og.zero_grad(set_to_none=False)
            im, la = gen(input=dr.z_pl, inject_index=dr.inject_pl, noise=dr.noise_pl, return_main_style_vectors=True)
            pn = dr.pl_image_noise / math.sqrt(im.shape[2] * im.shape[3] * im.shape[4])
            grads = torch.autograd.grad((im * pn).sum(), la, create_graph=True, retain_graph=True)[0]
            lpl, plen = pl_ref(grads)
            (hp["w_generator_regularization"] * lpl).backward()
            clip_step(gen, og, trace, "pl")
            log.update(path_length=plen.mean().item(), loss_pl=lpl.item())
            store[pre + "mean_path_length"] = npy(pl_ref.mean_path_length)
        with torch.no_grad():
            src = dict(gen.named_parameters())
            for n, p in gen_ema.named_parameters():
                before = p.detach().clone()
                p.mul_(0.999).add_(src[n], alpha=0.001)
                trace["ema.delta." + n] = p.detach() - before
        # ---- oracle
        otrace = {}
        olog = ot.train_iteration(ogen, odis, ogen_ema, oog, ood, pl_or, real, iteration, dr, trace=otrace,
                                  resume_training=late, top_k=top_k_or if late else None)
        for k, v in log.items():
            assert abs(olog[k] - v) <= 2e-4 * max(1.0, abs(v)), (k, olog[k], v)
            store[pre + "log." + k] = np.array(v)
        assert sorted(otrace) == sorted(trace)
        for k, v in trace.items():
            kind = k.split(".")[1]
            scale = max(v.abs().max().item(), 1e-30)
            err = (otrace[k] - v).abs().max().item() / scale
            if kind == "delta" and not k.startswith("ema."):
                # Adam(beta1=0) turns gradient rounding noise into +-lr: compare where the gradient is above noise
                gk = k.replace(".delta.", ".grad.", 1)
                mask = trace[gk].abs() > 0.05 * trace[gk].abs().max()
                err = ((otrace[k] - v).abs() * mask).max().item() / scale
            assert err <= (2e-3 if kind == "delta" else 5e-4), f"oracle != reference for {pre}{k}: {err:.3e}"
            if kind == "delta":          # movements: fp16 relative to the tensor's largest (5e-4 of it), 40 % of the file
                top = max(v.abs().max().item(), 1e-30)
                store[pre + "step." + k] = npy(v / top).astype(np.float16)
                store[pre + "step." + k.replace(".delta.", ".dscale.", 1)] = np.array(top, dtype=np.float64)
            else:
                store[pre + "step." + k] = npy(v)
        watch_g = ["style_mapping.layers.1.weight", "main_convolutions_1.5.modulated_convolution.weight",
                   "main_convolutions_1.2.modulated_convolution.modulation_mapping.bias",
                   "main_convolutions_2.3.modulated_convolution.weight", "output_blocks_2.0.modulated_convolution.weight"]
        watch_d = ["encoder_blocks.0.main_mapping.0.weight", "final_mapping.1.weight", "encoder_blocks.2.gamma"]
        for n in watch_g:
            close(dict(ogen.named_parameters())[n], dict(gen.named_parameters())[n], tol=1e-4, what="train.G." + n)
            store[pre + "G." + n] = npy(dict(gen.named_parameters())[n])
            store[pre + "Gema." + n] = npy(dict(gen_ema.named_parameters())[n])
        for n in watch_d:
            close(dict(odis.named_parameters())[n], dict(dis.named_parameters())[n], tol=1e-4, what="train.D." + n)
            store[pre + "D." + n] = npy(dict(dis.named_parameters())[n])


def gen_metrics(ref, store):
    """The statistics of multi_stylegan/validation_metrics.py that are callable without the pretrained networks: the static
    methods FID._calc_fid (:192-220) and FVD._calc_fvd (:401-429) on stored feature matrices, and misc.normalize_0_1_batch /
    normalize_m1_1_batch (misc.py:216-235).  The module imports torchvision and kornia at its top (for the networks and the
    resize): empty modules stand in, nothing of them is called here.  The inception-score arithmetic exists only inline in
    IS.__call__ (:126-140), which needs the Inception weights: not capturable, restated in oracle/metrics.py (unpinned)."""
    from oracle import metrics as omet
    sys.modules.setdefault("kornia", types.ModuleType("kornia"))
    vm = importlib.import_module("multi_stylegan.validation_metrics")
    rng = np.random.default_rng(7)
    cases = {"wide": (400, 24, 0.3), "few_samples": (20, 24, 1.0), "shifted": (300, 8, 2.0)}   # few_samples: rank-deficient covariances
    for name, (n, d, shift) in cases.items():
        mix_r, mix_f = rng.normal(size=(d, d)), rng.normal(size=(d, d))
        real = rng.normal(size=(n, d)) @ mix_r + rng.normal(size=d)
        fake = rng.normal(size=(n + 7, d)) @ mix_f + shift * rng.normal(size=d)
        fid, fvd = vm.FID._calc_fid(real, fake), vm.FVD._calc_fvd(real, fake)
        assert fid == fvd
        close(omet.frechet_distance(real, fake), fid, tol=1e-6, what=f"frechet distance {name}")   # (sqrtm: threaded Schur form, 1e-8 run to run)
        store[f"frechet.{name}.real"], store[f"frechet.{name}.fake"] = real, fake
        store[f"frechet.{name}.value"] = np.float64(fid)
    x = torch.randn(3, 3, 2, 5, 4, generator=torch.Generator().manual_seed(11)) * 3 + 1
    store["normalize.x"] = npy(x)
    store["normalize.y01"], store["normalize.ym11"] = npy(ref["misc"].normalize_0_1_batch(x)), npy(ref["misc"].normalize_m1_1_batch(x))


def gen_manifest(ref, om):
    G, D, C = ref["multi_stylegan_generator"], ref["u_net_2d_discriminator"], ref["config"]
    gen = G.Generator(C.multi_style_gan_generator_config)
    dis = D.Discriminator(C.u_net_2d_discriminator_config, no_rfp=True)
    man = {"generator": {k: list(v.shape) for k, v in gen.state_dict().items()},
           "discriminator": {k: list(v.shape) for k, v in dis.state_dict().items()},
           "generator_params": sum(p.numel() for p in gen.parameters()),
           "discriminator_params": sum(p.numel() for p in dis.parameters()),
           "generator_param_groups": [len(list(grp["params"])) for grp in gen.get_parameters()]}
    ogen, odis = om.Generator(), om.Discriminator(no_rfp=True)
    assert {k: list(v.shape) for k, v in ogen.state_dict().items()} == man["generator"]
    assert {k: list(v.shape) for k, v in odis.state_dict().items()} == man["discriminator"]
    assert [len(list(grp["params"])) for grp in ogen.get_parameters()] == man["generator_param_groups"]
    return man


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-only", action="store_true", help="compare oracle with reference, write nothing")
    args = ap.parse_args()
    torch.set_num_threads(8)
    ref = import_reference()
    from oracle import ops as oracle_ops, models as om, train as ot
    os.makedirs(OUT, exist_ok=True)
    files = {}
    for fname, fn in (("upfirdn2d", lambda s: gen_upfirdn(ref, oracle_ops, s)),
                      ("fused_act", lambda s: gen_fused_act(ref, oracle_ops, s)),
                      ("modconv", lambda s: gen_modconv(ref, om, s)),
                      ("layers", lambda s: gen_layers(ref, om, oracle_ops, s)),
                      ("train_step", lambda s: gen_train_step(ref, om, ot, s)),
                      ("metrics", lambda s: gen_metrics(ref, s))):
        store = {}
        fn(store)
        files[fname] = store
        print(f"{fname}: {len(store)} arrays, {sum(v.nbytes for v in store.values()) / 1e6:.2f} MB")
    store = {}
    extra, _ = gen_tiny_models(ref, om, store)
    files["tiny_models"] = store
    print(f"tiny_models: {len(store)} arrays, {sum(v.nbytes for v in store.values()) / 1e6:.2f} MB")
    manifest = gen_manifest(ref, om)
    manifest.update(extra)
    manifest["torch_version"] = torch.__version__
    if args.check_only:
        print("oracle agrees with the reference on every vector (nothing written)")
        return
    for fname, store in files.items():
        np.savez_compressed(os.path.join(OUT, fname + ".npz"), **store)
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
