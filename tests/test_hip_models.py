"""Whole-model and training-step parity on the GPU: product modules (multi_stylegan_amd) against golden vectors
captured from the reference's modules.  Tolerance 1e-3 relative (fp32 path), as BASELINE.json states."""
import copy
import json
import math
import os

import pytest
import torch

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3


def _models(golden, which="tiny_models", gp="tinyG.sd.", dp="tinyD.sd."):
    from tools.gen_golden import TINY_D, TINY_G
    import multi_stylegan_amd as m
    z = golden(which)
    g, d = m.MultiStyleGANGenerator(TINY_G), m.MultiStyleGANDiscriminator(TINY_D, no_rfp=True)
    g.load_state_dict(z.state_dict(gp)); d.load_state_dict(z.state_dict(dp))
    return z, g.to(DEV), d.to(DEV)


@pytest.mark.parametrize("elide", [False, True])
def test_tiny_generator(golden, elide):
    z, g, _ = _models(golden)
    g.elide_dead_branch = elide
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    zs = [z["tinyG.z0"].to(DEV), z["tinyG.z1"].to(DEV)]
    noise = [z[f"tinyG.noise{i}"].to(DEV) for i in range(7)]
    img, lat = g(zs, return_main_style_vectors=True, noise=noise, inject_index=3)
    assert img.shape == (3, 2, 3, 32, 32)
    assert rel_err(img, z["tinyG.image"]) < TOL and rel_err(lat, z["tinyG.latent"]) < TOL
    img.backward(z["tinyG.gimage"].to(DEV))
    none_grad = sorted(n for n, p in g.named_parameters() if p.grad is None)
    assert none_grad == man["tinyG.none_grad"]
    params = dict(g.named_parameters())
    for key in z.keys("tinyG.grad."):
        assert rel_err(params[key[len("tinyG.grad."):]].grad, z[key]) < TOL, key
    # path-length style double backward (create_graph through every custom op)
    g.zero_grad()
    im, la = g(zs, return_main_style_vectors=True, noise=noise, inject_index=3)
    gr, = torch.autograd.grad((im * z["tinyG.pl_image_noise"].to(DEV)).sum() / math.sqrt(3 * 32 * 32), la,
                              create_graph=True)
    pl = torch.sqrt(gr.pow(2).sum(2).mean(1) + 1e-8).mean()
    pl.backward()
    assert rel_err(gr, z["tinyG.pl_grads"]) < TOL and rel_err(pl, z["tinyG.pl"]) < TOL
    for key in z.keys("tinyG.plgrad."):
        assert rel_err(params[key[len("tinyG.plgrad."):]].grad, z[key]) < TOL, key


def test_generator_path_length_entry(golden):
    """Generator.forward(return_path_length_grads=True) with the image noise supplied."""
    z, g, _ = _models(golden)
    zs = [z["tinyG.z0"].to(DEV), z["tinyG.z1"].to(DEV)]
    noise = [z[f"tinyG.noise{i}"].to(DEV) for i in range(7)]
    gr = g(zs, noise=noise, inject_index=3, return_path_length_grads=True,
           path_length_noise=z["tinyG.pl_image_noise"].to(DEV))
    assert rel_err(gr, z["tinyG.pl_grads"]) < TOL


def test_tiny_discriminator(golden):
    z, _, d = _models(golden)
    x = z["tinyD.x"].to(DEV).requires_grad_(True)
    s, px = d(x)
    assert s.shape == (3, 1) and px.shape == (3, 1, 1, 32, 32)
    assert rel_err(s, z["tinyD.scalar"]) < TOL and rel_err(px, z["tinyD.pixel"]) < TOL
    gin, = torch.autograd.grad((s, px), x, (z["tinyD.gs"].to(DEV), z["tinyD.gpx"].to(DEV)), create_graph=True)
    assert rel_err(gin, z["tinyD.gin"]) < TOL
    r1 = 0.5 * gin.pow(2).reshape(3, -1).sum(1).mean()
    r1.backward()
    assert rel_err(r1, z["tinyD.r1"]) < TOL
    params = dict(d.named_parameters())
    for key in z.keys("tinyD.r1grad."):
        assert rel_err(params[key[len("tinyD.r1grad."):]].grad, z[key]) < TOL, key


def test_discriminator_concatenated_batches_equal_separate_calls(golden):
    """forward(cat([a, b]), minibatch_groups=2) == (forward(a), forward(b)): everything is per-sample except the
    minibatch statistic, which is taken per group; gradients of the summed loss agree as well."""
    z, _, d = _models(golden)
    torch.manual_seed(3)
    a, b = torch.rand(3, 2, 3, 32, 32, device=DEV), torch.rand(3, 2, 3, 32, 32, device=DEV)
    ca, pa = d(a)
    cb, pb = d(b)
    (ca.sum() + pa.sum() + 2 * cb.sum() + 2 * pb.sum()).backward()
    ref = [p.grad.clone() for p in d.parameters()]
    d.zero_grad()
    c2, p2 = d(torch.cat([a, b]), minibatch_groups=2)
    assert rel_err(c2, torch.cat([ca, cb])) < 1e-5 and rel_err(p2, torch.cat([pa, pb])) < 1e-5
    (c2[:3].sum() + p2[:3].sum() + 2 * c2[3:].sum() + 2 * p2[3:].sum()).backward()
    for p, r in zip(d.parameters(), ref):
        assert rel_err(p.grad, r) < 1e-4
    # one group over the concatenation is a DIFFERENT function (shared statistic)
    c1, _ = d(torch.cat([a, b]))
    assert rel_err(c1, torch.cat([ca, cb])) > 1e-6


def test_bf16_models_track_fp32(golden):
    """bf16 storage path: same graph, looser documented tolerance (5e-2 of max|ref|)."""
    z, g, d = _models(golden)
    g.compute_dtype = d.compute_dtype = torch.bfloat16
    zs = [z["tinyG.z0"].to(DEV), z["tinyG.z1"].to(DEV)]
    noise = [z[f"tinyG.noise{i}"].to(DEV) for i in range(7)]
    img = g(zs, noise=noise, inject_index=3)
    assert img.dtype == torch.float32 and rel_err(img, z["tinyG.image"]) < 5e-2
    s, px = d(z["tinyD.x"].to(DEV))
    assert rel_err(s, z["tinyD.scalar"]) < 5e-2 and rel_err(px, z["tinyD.pixel"]) < 5e-2


def test_train_iteration(golden):
    """ModelWrapper.train_iteration x2 (iterations 1 and 16 -> R1 and path length fire) vs the reference-driven
    golden run: four D losses, R1, G losses, path length + running mean, post-step parameters, EMA, and the dead
    second-stream weights staying bit-identical."""
    import multi_stylegan_amd as m
    from test_oracle_golden import load_train_draws
    z, g, d = _models(golden, "train_step", "train.G0.", "train.D0.")
    dead0 = g.main_convolutions_2[3].modulated_convolution.weight.detach().clone()
    trainer = m.ModelWrapper(g, d, device=DEV)
    names = {"loss_d_real": "loss_discriminator_real", "loss_d_fake": "loss_discriminator_fake",
             "loss_d_real_px": "loss_discriminator_real_pixel_wise", "loss_d_fake_px": "loss_discriminator_fake_pixel_wise",
             "r1": "loss_discriminator_regularization", "loss_g": "loss_generator",
             "loss_g_px": "loss_generator_pixel_wise", "path_length": "path_length",
             "loss_pl": "loss_path_length_regularization"}
    for step, iteration in enumerate((1, 16)):
        real, draws = load_train_draws(z, step, m.model_wrapper)
        trainer.iteration = iteration - 1
        trainer.train_iteration(real.to(DEV), draws.to(DEV))
        log = trainer.pop_logs()
        pre = f"train.it{step}."
        for key in z.keys(pre + "log."):
            want, got = float(z[key]), log[names[key[len(pre + "log."):]]][0]
            assert abs(got - want) <= TOL * max(1.0, abs(want)), (key, got, want)
        gp, dp = dict(g.named_parameters()), dict(d.named_parameters())
        ep = dict(trainer.generator_ema.named_parameters())
        for key in z.keys(pre + "G."):
            assert rel_err(gp[key[len(pre + "G."):]], z[key]) < TOL, key
        for key in z.keys(pre + "Gema."):
            assert rel_err(ep[key[len(pre + "Gema."):]], z[key]) < TOL, key
        for key in z.keys(pre + "D."):
            assert rel_err(dp[key[len(pre + "D."):]], z[key]) < TOL, key
    assert rel_err(trainer.path_length_regularization.mean_path_length, z["train.it1.mean_path_length"]) < TOL
    assert torch.equal(g.main_convolutions_2[3].modulated_convolution.weight.cpu(), dead0.cpu())


def test_config1_64px_matches_oracle():
    """BASELINE config 1 (64x64, 5 stages x 512 channels, B=4): product on the GPU vs the CPU oracle with the
    same weights, z and noise."""
    import multi_stylegan_amd as m
    from multi_stylegan_amd.config import generator_config_for_resolution
    from oracle import models as om
    torch.manual_seed(3)
    cfg = generator_config_for_resolution(64)
    go = om.Generator(cfg)
    gd = m.MultiStyleGANGenerator(cfg)
    gd.load_state_dict(go.state_dict())
    gd.to(DEV)
    z = [torch.randn(4, 512), torch.randn(4, 512)]
    noise = [torch.randn(4, 1, 4, 4)] + [torch.randn(4, 1, 2 ** (i // 2 + 3), 2 ** (i // 2 + 3)) for i in range(8)]
    with torch.no_grad():
        want = go(z, noise=noise, inject_index=4)
        got = gd([t.to(DEV) for t in z], noise=[t.to(DEV) for t in noise], inject_index=4)
    assert rel_err(got, want) < TOL
    do = om.Discriminator(no_rfp=True)
    dd = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
    dd.load_state_dict(do.state_dict())
    dd.to(DEV)
    x = torch.rand(4, 2, 3, 64, 64)
    with torch.no_grad():
        ws, wp = do(x)
        gs, gp = dd(x.to(DEV))
    assert rel_err(gs, ws) < TOL and rel_err(gp, wp) < TOL


@pytest.mark.parametrize("batch", [1, 3])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_train_iteration_odd_batches(golden, batch, dtype):
    """Batch sizes the tile / group logic could trip over (1: every minibatch statistic degenerates; 3: the path-length
    half batch is 1, nothing divides anything): two iterations incl. the regularised 16th run and stay finite."""
    import multi_stylegan_amd as m
    _, g, d = _models(golden)
    g.compute_dtype = d.compute_dtype = dtype
    tr = m.ModelWrapper(g, d, device=DEV)
    tr.generator_ema.compute_dtype = dtype
    tr.iteration = 14
    torch.manual_seed(batch)
    for _ in range(2):
        tr.train_iteration(torch.rand(batch, 2, 3, 32, 32, device=DEV))
    logs = tr.pop_logs()
    assert {"loss_discriminator_regularization", "path_length", "loss_generator"} <= set(logs)
    assert all(math.isfinite(v) for vals in logs.values() for v in vals)
    assert all(torch.isfinite(p).all() for p in list(g.parameters()) + list(d.parameters()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_block_input_gradient_merge(dtype):
    """A ResNet block's input feeds the main 3x3 conv and the 1x1 residual conv; first-order backward hands the
    residual conv's input gradient to the 3x3 conv's data-gradient epilogue (conv_ops.fork_input).  Same gradients as
    autograd's own accumulation, the hand-over really happens, and second-order graphs (R1) fall back to the plain add."""
    import multi_stylegan_amd as m
    from multi_stylegan_amd import conv_ops, u_net_2d_discriminator as U
    torch.manual_seed(5)
    block = U.ResNetBlock(64, 32).to(DEV)
    x0 = conv_ops.to_compute_layout(torch.randn(3, 64, 24, 20, device=DEV), dtype)
    gy = conv_ops.to_compute_layout(torch.randn(3, 32, 24, 20, device=DEV), dtype)
    calls = []
    orig = conv_ops._d_raw
    def spy(gy_, w_, g_, residual=None):
        calls.append(residual is not None)
        return orig(gy_, w_, g_, residual=residual)
    grads = {}
    for flag in (False, True):
        U.FUSE_INPUT_FORK = flag
        calls.clear()
        conv_ops._d_raw = spy
        try:
            x = x0.clone().requires_grad_(True)
            block.zero_grad()
            block(x).backward(gy)
        finally:
            conv_ops._d_raw = orig
        grads[flag] = (x.grad.float(), [p.grad.clone() for p in block.parameters()])
        assert any(calls) == flag, "the merged data-gradient epilogue must run exactly when the fork is enabled"
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(grads[True][0], grads[False][0]) < tol
    for a, b in zip(grads[True][1], grads[False][1]):
        assert rel_err(a, b) < tol
    # second order: grad of |dy/dx|^2 wrt the weights, merged path must step aside
    U.FUSE_INPUT_FORK = True
    outs = {}
    for flag in (False, True):
        U.FUSE_INPUT_FORK = flag
        x = x0.clone().requires_grad_(True)
        gx, = torch.autograd.grad(block(x).float().sum(), x, create_graph=True)
        pen = gx.float().square().sum()
        outs[flag] = torch.autograd.grad(pen, list(block.parameters()))
    U.FUSE_INPUT_FORK = True
    for a, b in zip(outs[True], outs[False]):
        assert rel_err(a, b) < (1e-4 if dtype == torch.float32 else 5e-2)


def test_full_size_models_bf16_track_fp32():
    """The production shapes (256^2, 512 channels) through the kernels only they reach -- the row-sharing 3x3 kernels,
    the ping-pong kernel, the sub-pixel up-conv -- in bf16 storage, against the fp32-storage path of the same models
    (different kernels: exact-fp32 MFMA, 128x128 tiles): forward values and the gradients of one backward."""
    import multi_stylegan_amd as m
    from multi_stylegan_amd.config import generator_config_for_resolution
    torch.manual_seed(7)
    gen = m.MultiStyleGANGenerator(generator_config_for_resolution(256)).to(DEV)
    dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True).to(DEV)
    z = [torch.randn(2, 512, device=DEV), torch.randn(2, 512, device=DEV)]
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        gen.compute_dtype = dis.compute_dtype = dt
        gen.zero_grad(); dis.zero_grad()
        torch.manual_seed(11)                       # the same per-layer noise draws in both passes
        img = gen(z, inject_index=5)
        score, pixel = dis(img)
        (score.mean() + pixel.mean() + img.mean()).backward()
        res[dt] = (img.detach(), score.detach(), pixel.detach(),
                   gen.main_convolutions_1[11].modulated_convolution.weight.grad.clone(),
                   gen.style_mapping.layers[1].weight.grad.clone(),
                   dis.encoder_blocks[0].main_mapping[0].weight.grad.clone())
    for name, a, b in zip(("image", "score", "pixel map", "G conv weight grad", "G mapping weight grad", "D conv weight grad"),
                          res[torch.bfloat16], res[torch.float32]):
        assert torch.isfinite(a).all(), name
        err = ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()
        print(f"{name}: {err:.4f}")
        # forward values: bf16 storage drift through ~30 layers; gradients additionally see leaky-ReLU slopes flip
        # where a pre-activation is within rounding of zero
        assert err < (6e-2 if "grad" not in name else 0.2), (name, err)
