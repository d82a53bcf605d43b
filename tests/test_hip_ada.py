"""Adaptive discriminator augmentation on the GPU against oracle/ada.py on identical draws.  Parity with
the reference itself is UNPINNED for this path (kornia 0.4.1 is not available): the oracle restates kornia's published
algorithm through torch's own affine_grid / grid_sample."""
import copy
import math
import random

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import ada as oa

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _to_product_draws(dr: oa.Draws, device):
    n = dr.u_flip.shape[0]
    return {"u": torch.stack([dr.u_flip, dr.u_rot90, dr.u_roll, dr.u_iso, dr.u_rot_a, dr.u_aniso, dr.u_rot_b]).to(device),
            "angle90": dr.angle90, "roll": dr.roll,
            "scale_iso": dr.scale_iso.view(n, 1).expand(n, 2).contiguous().to(device),
            "angle_a": dr.angle_a.to(device), "scale_aniso": dr.scale_aniso.to(device), "angle_b": dr.angle_b.to(device)}


def _seed(s):
    torch.manual_seed(s); random.seed(s); np.random.seed(s)


@pytest.mark.parametrize("p", [0.0, 0.3, 0.8, 1.0])
@pytest.mark.parametrize("shape", [(6, 6, 40, 40), (3, 6, 64, 48), (16, 6, 256, 256)])
def test_pipeline_matches_oracle(shape, p):
    from multi_stylegan_amd import AugmentationPipeline
    _seed(sum(shape) + int(10 * p))
    n, c, h, w = shape
    for trial in range(3 if h < 256 else 1):
        images = torch.rand(shape)
        dr = oa.draw(n, h, w)
        if trial == 1:
            dr.angle90 = 90.0                      # every multiple of 90 degrees gets exercised
        if trial == 2:
            dr.angle90 = 180.0
        x_ref = images.clone().requires_grad_(True)
        want = oa.augment(x_ref, p, dr)
        gy = torch.randn(shape)
        want.backward(gy)
        x = images.to(DEV).requires_grad_(True)
        got = AugmentationPipeline()(x, torch.tensor(p, device=DEV), _to_product_draws(dr, DEV))
        got.backward(gy.to(DEV))
        # bilinear sampling of images in [0, 1]: coordinates agree to ~1e-5 pixel
        assert rel_err(got, want) < 2e-4, (trial, rel_err(got, want))
        assert rel_err(x.grad, x_ref.grad) < 5e-4, (trial, rel_err(x.grad, x_ref.grad))
        if p == 0.0:
            assert torch.equal(got.cpu(), images)


def test_pipeline_draws_its_own_randomness():
    """Without explicit draws: finite output of the same shape, p as a device scalar, reproducible under the reference's
    three RNG sources (torch, random, numpy)."""
    from multi_stylegan_amd import AugmentationPipeline
    images = torch.rand(8, 6, 32, 32, device=DEV)
    outs = []
    for _ in range(2):
        _seed(5)
        outs.append(AugmentationPipeline()(images, torch.tensor(0.6, device=DEV)))
    assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0]).all() and not torch.equal(outs[0], images)


class _FixedDiscriminator(torch.nn.Module):
    """Returns preset predictions (the controller only looks at their signs)."""

    def __init__(self, outputs):
        super().__init__()
        self.outputs = list(outputs)
        self.compute_dtype = torch.float32

    def forward(self, images, **kwargs):
        return self.outputs.pop(0)


def test_controller_matches_oracle_and_stays_on_device():
    """p trajectory over 80 fake and 80 real batches against the reference's host-side controller (:76-94), including
    the clamps at 0 and p_max; real batches do not count; cut-mix calls bypass augmentation and controller."""
    from multi_stylegan_amd import AdaptiveDiscriminatorAugmentation
    _seed(1)
    preds = []
    for i in range(160):
        bias = 1.5 if i < 64 else -1.5          # D very sure the fakes are ... real, then the opposite
        preds.append((torch.randn(4, 1) + bias, torch.randn(4, 1, 1, 8, 8) + bias))
    ref = oa.Controller(p_step=0.05, r_update=4, p_max=0.3)
    ada = AdaptiveDiscriminatorAugmentation(_FixedDiscriminator([(a.to(DEV), b.to(DEV)) for a, b in preds]),
                                            p_step=0.05, r_update=4, p_max=0.3)
    images = torch.rand(4, 2, 3, 8, 8, device=DEV)
    seen = []
    for i, (a, b) in enumerate(preds):
        is_real = i % 2 == 1
        ada(images.clone(), is_real=is_real)
        ref.observe(a, b, is_real)
        seen.append((ada.p, ref.p))
    assert all(abs(x - y) < 1e-6 for x, y in seen), seen[:12]
    assert max(x for x, _ in seen) == pytest.approx(0.3) and min(x for x, _ in seen) == pytest.approx(0.0, abs=1e-7)
    assert len(ada.r_history) == len(ref.r_history)
    state = ada.ada_state()
    other = AdaptiveDiscriminatorAugmentation(_FixedDiscriminator([]))
    other.load_ada_state(state)
    assert other.p == pytest.approx(ada.p) and other._r_count == ada._r_count


def test_wrapper_augments_in_place_and_skips_cut_mix():
    from multi_stylegan_amd import AdaptiveDiscriminatorAugmentation
    _seed(2)
    out = (torch.zeros(4, 1, device=DEV), torch.zeros(4, 1, 1, 16, 16, device=DEV))
    ada = AdaptiveDiscriminatorAugmentation(_FixedDiscriminator([out, out, out]))
    ada.p = 1.0
    images = torch.rand(4, 2, 3, 16, 16, device=DEV)
    before = images.clone()
    ada(images, is_real=True)
    assert not torch.equal(images, before)        # the caller's batch now holds the augmented images (reference :64-68)
    before = images.clone()
    ada(images, is_cut_mix=True)
    assert torch.equal(images, before)
    leaf = torch.rand(4, 2, 3, 16, 16, device=DEV, requires_grad=True)
    kept = leaf.detach().clone()
    ada(leaf, is_real=False)
    assert torch.equal(leaf.detach(), kept)       # a batch that carries gradients is augmented out of place


def test_trainer_with_ada(golden, tmp_path):
    """Config-5 wiring: the trainer with an ADA-wrapped discriminator runs iterations 15-17 (R1 and path length
    included), p stays in range, its state travels through a checkpoint, the reference-layout discriminator keys carry
    the wrapper's `discriminator.` prefix."""
    import multi_stylegan_amd as m
    from tools.gen_golden import TINY_D, TINY_G
    _seed(3)
    z = golden("tiny_models")
    g, d = m.MultiStyleGANGenerator(TINY_G), m.MultiStyleGANDiscriminator(TINY_D, no_rfp=True)
    g.load_state_dict(z.state_dict("tinyG.sd.")); d.load_state_dict(z.state_dict("tinyD.sd."))
    ada = m.AdaptiveDiscriminatorAugmentation(d, r_update=2)
    tr = m.ModelWrapper(g, ada, device=DEV)
    assert tr.batch_discriminator_passes          # real + fake as one batch through the wrapper
    tr.iteration = 14
    for _ in range(3):
        tr.train_iteration(torch.rand(4, 2, 3, 32, 32, device=DEV))
    logs = tr.pop_logs()
    assert all(math.isfinite(v) for vals in logs.values() for v in vals)
    assert {"loss_discriminator_regularization", "path_length"} <= set(logs)
    assert 0.0 <= ada.p <= 0.8 and len(ada.r_history) == 3        # 6 fake batches (D step + G step per iteration) / 2
    path = str(tmp_path / "ck.pt")
    tr.save_checkpoint(path)
    ck = torch.load(path, weights_only=False)
    assert all(k.startswith("discriminator.") for k in ck["discriminator"])
    assert ck["multi_stylegan_amd"]["ada"]["p"] == pytest.approx(ada.p)
    d2 = m.MultiStyleGANDiscriminator(TINY_D, no_rfp=True)
    tr2 = m.ModelWrapper(copy.deepcopy(g), m.AdaptiveDiscriminatorAugmentation(d2), device=DEV)
    tr2.load_checkpoint(path)
    assert tr2.discriminator.p == pytest.approx(ada.p)
    # ... and into a trainer whose discriminator is NOT wrapped (prefix stripped)
    d3 = m.MultiStyleGANDiscriminator(TINY_D, no_rfp=True)
    tr3 = m.ModelWrapper(copy.deepcopy(g), d3, device=DEV)
    tr3.load_checkpoint(path)
    for (n, a), (_, b) in zip(d.state_dict().items(), d3.state_dict().items()):
        assert torch.equal(a, b), n


@pytest.mark.parametrize("padding,align", [(2, True), (0, False), (0, True), (2, False)])
def test_affine_warp_random_geometry(padding, align):
    """msg_affine_warp one stage at a time against oracle.ada.warp_affine (torch affine_grid + grid_sample) over random
    non-square images, centres, angles, anisotropic scales from 0.4 to 2.5, with every image selected and with a random
    subset selected; forward and the gradient of a random cotangent."""
    from multi_stylegan_amd.adaptive_discriminator_augmentation import affine_warp
    gen = torch.Generator().manual_seed(7 + padding + int(align))
    mode = "reflection" if padding == 2 else "zeros"
    for trial in range(6):
        n, c = int(torch.randint(1, 7, (1,), generator=gen)), int(torch.randint(1, 7, (1,), generator=gen))
        h, w = int(torch.randint(5, 40, (1,), generator=gen)), int(torch.randint(5, 40, (1,), generator=gen))
        x = torch.rand(n, c, h, w, generator=gen)
        angle = (torch.rand(n, generator=gen) - 0.5) * 360.0
        scale = 0.4 + 2.1 * torch.rand(n, 2, generator=gen)
        center = (float(torch.rand(1, generator=gen)) * w, float(torch.rand(1, generator=gen)) * h)
        u = torch.rand(n, generator=gen)
        p = 1.0 if trial % 2 == 0 else 0.5
        sel = u <= p
        m = oa.rotation_matrix2d(torch.tensor([center]).expand(n, 2), angle, scale)
        xr = x.clone().requires_grad_(True)
        warped = oa.warp_affine(xr, m, mode, align)
        want = torch.where(sel.view(n, 1, 1, 1), warped, xr)
        gy = torch.randn(want.shape, generator=gen)
        want.backward(gy)
        xd = x.to(DEV).requires_grad_(True)
        got = affine_warp(xd, u.to(DEV), torch.tensor(p, device=DEV), angle=angle.to(DEV), scale=scale.to(DEV),
                          center=center, padding=padding, align_corners=align)
        got.backward(gy.to(DEV))
        assert rel_err(got, want) < 5e-4, (trial, n, c, h, w, rel_err(got, want))
        assert rel_err(xd.grad, xr.grad) < 2e-3, (trial, rel_err(xd.grad, xr.grad))


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), -float("inf"), 1e9])
def test_affine_warp_backward_keeps_a_blow_up_visible(bad):
    """The scatter accumulates in 64-bit fixed point; a non-finite or out-of-range cotangent has no image there and used to
    come out as finite garbage.  It must surface as NaN in the gradient of the transformed images (so finite checks and the
    norm clipping see it) and leave copied-through images untouched."""
    from multi_stylegan_amd.adaptive_discriminator_augmentation import affine_warp
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(3, 2, 12, 10, generator=gen).to(DEV).requires_grad_(True)
    u = torch.tensor([0.1, 0.9, 0.2], device=DEV)                        # image 1 is copied through at p = 0.5
    y = affine_warp(x, u, torch.tensor(0.5, device=DEV), angle=torch.tensor([20.0, 0.0, -35.0], device=DEV),
                    scale=torch.ones(3, 2, device=DEV), center=(4.5, 5.5), padding=2, align_corners=True)
    gy = torch.randn(y.shape, generator=gen).to(DEV)
    gx_ok, = torch.autograd.grad(y, x, gy, retain_graph=True)
    assert torch.isfinite(gx_ok).all()
    gy_bad = gy.clone()
    gy_bad[0, 1, 5, 5] = bad
    gx, = torch.autograd.grad(y, x, gy_bad)
    assert not torch.isfinite(gx[0]).all() and not torch.isfinite(gx[2]).all()
    assert torch.equal(gx[1], gy_bad[1])                                  # the pass-through image: the cotangent itself


def test_wrapper_pair_forward_equals_two_calls(golden):
    """forward(cat([real, fake]), minibatch_groups=2) == forward(real, is_real=True) then forward(fake): same
    predictions on the same draws, the controller fed by the fake half only, both halves rewritten in place."""
    import multi_stylegan_amd as m
    from tools.gen_golden import TINY_D
    _seed(9)
    z = golden("tiny_models")
    d = m.MultiStyleGANDiscriminator(TINY_D, no_rfp=True)
    d.load_state_dict(z.state_dict("tinyD.sd."))
    d.to(DEV)
    real, fake = torch.rand(3, 2, 3, 32, 32, device=DEV), torch.rand(3, 2, 3, 32, 32, device=DEV)
    dr_r, dr_f = (_to_product_draws(oa.draw(3, 32, 32), DEV) for _ in range(2))
    a1 = m.AdaptiveDiscriminatorAugmentation(d, r_update=1)
    a1.p = 0.7
    r1, f1 = real.clone(), fake.clone()
    sr, pr = a1(r1, is_real=True, draws=dr_r)
    sf, pf = a1(f1, is_real=False, draws=dr_f)
    a2 = m.AdaptiveDiscriminatorAugmentation(d, r_update=1)
    a2.p = 0.7
    both = torch.cat([real, fake])
    s2, p2 = a2(both, minibatch_groups=2, draws=(dr_r, dr_f))
    assert rel_err(s2, torch.cat([sr, sf])) < 1e-5 and rel_err(p2, torch.cat([pr, pf])) < 1e-5
    assert torch.equal(both[:3], r1) and torch.equal(both[3:], f1) and not torch.equal(r1, real)
    assert a1.p == pytest.approx(a2.p) and len(a1.r_history) == len(a2.r_history) == 1


def test_config5_full_size_training_iteration_with_ada():
    """BASELINE config 5's per-GPU work as ONE training iteration at its own size: 256x256, 7 x 512 channels, batch 16, bf16
    storage, the discriminator (non-local blocks are always on) wrapped in adaptive discriminator augmentation
    (reference adaptive_discriminator_augmentation.py:63-96).  Iterations 15 (plain) and 16 (R1 on the batch ADA augmented in
    place -- the trainer's concatenated real + fake batch, test_wrapper_augments_in_place_and_skips_cut_mix --, path length):
    every loss finite, the controller counted the four fake
    batches it saw (D step + G step per iteration) and moved p by +-p_step on the DEVICE (no host round trip: `_p` is a device
    tensor, `r_history` holds device scalars), parameters finite, memory well inside one MI355X."""
    import multi_stylegan_amd as m
    from multi_stylegan_amd.config import generator_config_for_resolution
    _seed(51)
    g = m.MultiStyleGANGenerator(generator_config_for_resolution(256))
    d = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
    g.compute_dtype = d.compute_dtype = torch.bfloat16
    ada = m.AdaptiveDiscriminatorAugmentation(d, r_update=4)
    ada.p = 0.5                                          # half of the images take every stage of the pipeline
    tr = m.ModelWrapper(g, ada, device=DEV)
    tr.generator_ema.compute_dtype = torch.bfloat16
    assert tr.batch_discriminator_passes
    tr.iteration = 14
    torch.cuda.reset_peak_memory_stats()
    real = torch.rand(16, 2, 3, 256, 256, device=DEV)
    for it in (15, 16):
        tr.train_iteration(real.clone())
    logs = tr.pop_logs()
    assert {"loss_discriminator_real", "loss_discriminator_regularization", "loss_generator", "path_length"} <= set(logs)
    assert all(math.isfinite(v) for vals in logs.values() for v in vals), logs
    assert ada._p.is_cuda and len(ada.r_history) == 1 and ada.r_history[0].is_cuda and ada._r_count == 0
    assert abs(abs(ada.p - 0.5) - ada.p_step) < 1e-6, ada.p           # one controller update: p moved by exactly one step
    assert -1.0 <= float(ada.r_history[0]) <= 1.0
    assert all(torch.isfinite(p).all() for p in list(g.parameters()) + list(d.parameters()))
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    print(f"config 5 per-GPU iteration pair: p 0.5 -> {ada.p:.3f}, r {float(ada.r_history[0]):+.3f}, peak {peak:.1f} GiB, "
          f"losses " + " ".join(f"{k}={v[-1]:.4g}" for k, v in logs.items()))
    assert peak < 40.0
