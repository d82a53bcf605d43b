"""The oracle (oracle/) against the golden vectors captured from the reference's own modules.

These pin the oracle; they run on CPU.  Tolerance: fp32 round-off only (2e-5 relative to max|ref|).
"""
import json
import math
import os

import pytest
import torch

from conftest import GOLDEN, check_step_trace, rel_err
from oracle import models as om
from oracle import ops as oo
from oracle import train as ot

TOL = 2e-5
UPFIRDN_CASES = ["g_blur_pad21_gain4", "g_skip_up2_pad21", "d_blur_pad22_odd", "bwd_of_up2_down2",
                 "asym_blur_pad21", "asym_up2_pad21", "asym_down2_pad12", "blur_pad11"]


@pytest.mark.parametrize("case", UPFIRDN_CASES)
def test_upfirdn2d(golden, case):
    z = golden("upfirdn2d")
    up, down, p0, p1 = [int(v) for v in z[case + ".cfg"]]
    x = z[case + ".x"].requires_grad_(True)
    gy = z[case + ".gy"].requires_grad_(True)
    y = oo.upfirdn2d(x, z[case + ".fir"], up=up, down=down, pad=(p0, p1))
    gx, = torch.autograd.grad(y, x, gy, create_graph=True)
    ggy, = torch.autograd.grad(gx, gy, z[case + ".ggx"])
    assert rel_err(y, z[case + ".y"]) < TOL
    assert rel_err(gx, z[case + ".gx"]) < TOL
    assert rel_err(ggy, z[case + ".ggy"]) < TOL
    lit = oo.upfirdn2d_scalar(x.detach().reshape(-1, *x.shape[2:]), z[case + ".fir"], up, down, p0, p1)
    assert rel_err(lit.reshape(y.shape), z[case + ".y"]) < TOL


@pytest.mark.parametrize("case", ["mlp_2d", "conv_4d", "conv_4d_sqrt2"])
def test_fused_leaky_relu(golden, case):
    z = golden("fused_act")
    x, b = z[case + ".x"].requires_grad_(True), z[case + ".b"].requires_grad_(True)
    gy = z[case + ".gy"].requires_grad_(True)
    y = oo.fused_leaky_relu(x, b, 0.2, float(z[case + ".scale"]))
    gx, gb = torch.autograd.grad(y, (x, b), gy, create_graph=True)
    ggy, = torch.autograd.grad((gx, gb), gy, (z[case + ".ggx"], z[case + ".ggb"]))
    for got, key in ((y, "y"), (gx, "gx"), (gb, "gb"), (ggy, "ggy")):
        assert rel_err(got, z[f"{case}.{key}"]) < TOL, key


MODCONV = {"conv3x3_demod": dict(kernel_size=(3, 3), demodulate=True, upsampling=False),
           "up2x2_demod": dict(kernel_size=(2, 2), demodulate=True, upsampling=True),
           "torgb1x1_nodemod": dict(kernel_size=(1, 1), demodulate=False, upsampling=False)}


@pytest.mark.parametrize("kind", list(MODCONV))
@pytest.mark.parametrize("mapped", [True, False])
def test_modulated_conv(golden, kind, mapped):
    z = golden("modconv")
    name = f"{kind}.{'map' if mapped else 'nomap'}"
    out_c = 3 if kind.startswith("torgb") else 12
    m = om.ModulatedConv2d(8, out_c, 10, modulation_mapping=mapped, **MODCONV[kind])
    m.load_state_dict(z.state_dict(name + ".sd."))
    x, st = z[name + ".x"].requires_grad_(True), z[name + ".style"].requires_grad_(True)
    res = m(x, st)
    y = res[0] if mapped else res
    gx, gst, gw = torch.autograd.grad(y, (x, st, m.weight), z[name + ".gy"], create_graph=True)
    gg, = torch.autograd.grad(gx.square().sum(), st)
    assert rel_err(y, z[name + ".y"]) < TOL
    assert rel_err(gx, z[name + ".gx"]) < TOL
    assert rel_err(gst, z[name + ".gstyle"]) < TOL
    assert rel_err(gw, z[name + ".gweight"]) < TOL
    assert rel_err(gg, z[name + ".gg_style"]) < 1e-4
    if mapped:
        assert rel_err(res[1], z[name + ".style_out"]) < TOL


def test_small_layers(golden):
    z = golden("layers")
    assert rel_err(oo.equalized_linear(z["eqlinear.x"], z["eqlinear.w"], z["eqlinear.b"]), z["eqlinear.y"]) < TOL
    for name in ("eqconv3x3", "eqconv3x3_s2_bias", "eqconv1x1"):
        stride, pad = [int(v) for v in z[name + ".cfg"]]
        b = z[name + ".b"] if name + ".b" in z.keys() else None
        x, w = z[name + ".x"].requires_grad_(True), z[name + ".w"].requires_grad_(True)
        y = oo.equalized_conv2d(x, w, b, stride, pad)
        gx, gw = torch.autograd.grad(y, (x, w), z[name + ".gy"])
        assert rel_err(y, z[name + ".y"]) < TOL and rel_err(gx, z[name + ".gx"]) < TOL
        assert rel_err(gw, z[name + ".gw"]) < TOL
    # the two layers of the public surface that the models do not instantiate (equalized_layer.py:77-207)
    for name, fn in (("eqconvT2x2", oo.equalized_conv_transpose2d), ("eqconv1d", oo.equalized_conv1d),
                     ("eqconv1d_s2", lambda a, w, b: oo.equalized_conv1d(a, w, b, stride=2, padding=2))):
        x, w = z[name + ".x"].requires_grad_(True), z[name + ".w"].requires_grad_(True)
        y = fn(x, w, z[name + ".b"])
        gx, gw = torch.autograd.grad(y, (x, w), z[name + ".gy"])
        assert rel_err(y, z[name + ".y"]) < TOL and rel_err(gx, z[name + ".gx"]) < TOL and rel_err(gw, z[name + ".gw"]) < TOL
    assert rel_err(oo.pixel_norm(z["pixelnorm.x"]), z["pixelnorm.y"]) < TOL
    assert rel_err(oo.minibatch_stddev(z["mbstd.x"]), z["mbstd.y"]) < TOL
    for name, blk in (("nonlocal", om.NonLocalBlock(8, 16)), ("resnet_mbstd", om.ResNetBlock(8, 12, True))):
        blk.load_state_dict(z.state_dict(name + ".sd."))
        x = z[name + ".x"].requires_grad_(True)
        y = blk(x)
        gx, = torch.autograd.grad(y, x, z[name + ".gy"])
        assert rel_err(y, z[name + ".y"]) < TOL and rel_err(gx, z[name + ".gx"]) < TOL


def _tiny(golden):
    from tools.gen_golden import TINY_D, TINY_G
    z = golden("tiny_models")
    g, d = om.Generator(TINY_G), om.Discriminator(TINY_D, no_rfp=True)
    g.load_state_dict(z.state_dict("tinyG.sd.")); d.load_state_dict(z.state_dict("tinyD.sd."))
    return z, g, d


def test_tiny_generator(golden):
    z, g, _ = _tiny(golden)
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    zs = [z["tinyG.z0"], z["tinyG.z1"]]
    noise = [z[f"tinyG.noise{i}"] for i in range(7)]
    img, lat = g(zs, return_main_style_vectors=True, noise=noise, inject_index=3)
    assert rel_err(img, z["tinyG.image"]) < TOL and rel_err(lat, z["tinyG.latent"]) < TOL
    img.backward(z["tinyG.gimage"])
    none_grad = sorted(n for n, p in g.named_parameters() if p.grad is None)
    assert none_grad == man["tinyG.none_grad"]          # quirk Q1: the dead second stream
    params = dict(g.named_parameters())
    for key in z.keys("tinyG.grad."):
        assert rel_err(params[key[len("tinyG.grad."):]].grad, z[key]) < 1e-4, key
    g.zero_grad()
    im, la = g(zs, return_main_style_vectors=True, noise=noise, inject_index=3)
    gr, = torch.autograd.grad((im * z["tinyG.pl_image_noise"]).sum() / math.sqrt(3 * 32 * 32), la, create_graph=True)
    pl = torch.sqrt(gr.pow(2).sum(2).mean(1) + 1e-8).mean()
    pl.backward()
    assert rel_err(gr, z["tinyG.pl_grads"]) < 1e-4 and rel_err(pl, z["tinyG.pl"]) < 1e-4
    for key in z.keys("tinyG.plgrad."):
        assert rel_err(params[key[len("tinyG.plgrad."):]].grad, z[key]) < 2e-4, key


def test_tiny_discriminator(golden):
    z, _, d = _tiny(golden)
    x = z["tinyD.x"].requires_grad_(True)
    s, px = d(x)
    assert rel_err(s, z["tinyD.scalar"]) < TOL and rel_err(px, z["tinyD.pixel"]) < TOL
    gin, = torch.autograd.grad((s, px), x, (z["tinyD.gs"], z["tinyD.gpx"]), create_graph=True)
    assert rel_err(gin, z["tinyD.gin"]) < 1e-4
    r1 = 0.5 * gin.pow(2).reshape(3, -1).sum(1).mean()
    r1.backward()
    assert rel_err(r1, z["tinyD.r1"]) < 1e-4
    params = dict(d.named_parameters())
    for key in z.keys("tinyD.r1grad."):
        assert rel_err(params[key[len("tinyD.r1grad."):]].grad, z[key]) < 2e-4, key


def test_state_dict_manifest():
    """API parity: key/shape set of the default 256^2 models equals the reference's (201 / 72 entries)."""
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    g, d = om.Generator(), om.Discriminator(no_rfp=True)
    assert {k: list(v.shape) for k, v in g.state_dict().items()} == man["generator"]
    assert {k: list(v.shape) for k, v in d.state_dict().items()} == man["discriminator"]
    assert len(man["generator"]) == 201 and len(man["discriminator"]) == 72
    assert sum(p.numel() for p in g.parameters()) == man["generator_params"] == 53018664
    assert sum(p.numel() for p in d.parameters()) == man["discriminator_params"] == 50855682


GOLDEN_ITERATIONS = ((1, False), (16, False), (32, True))      # (iteration, late-training branches on)
STEP_LABELS = {1: ["d", "g"], 16: ["d", "r1", "g", "pl"], 32: ["d", "r1", "cm_aug", "cm_reg", "g", "pl"]}


def load_train_draws(z, step, ot_mod=ot):
    pre = f"train.it{step}."
    def lst(key):
        return [z[k] for k in sorted(z.keys(pre + key + "."), key=lambda s: int(s.rsplit(".", 1)[1]))]
    zg = lst("z_g")
    late = bool(z.keys(pre + "wrong_order_perm"))
    extra = dict(wrong_order_perm=z[pre + "wrong_order_perm"], cut_mix=True, cut_mix_map_aug=z[pre + "cut_mix_map_aug"],
                 cut_mix_map_reg=z[pre + "cut_mix_map_reg"]) if late else dict(cut_mix=False)
    return z[pre + "real"], ot_mod.Draws(
        z_d=lst("z_d"), inject_d=2, noise_d=lst("noise_d"), z_g=zg[0] if len(zg) == 1 else zg, noise_g=lst("noise_g"),
        z_pl=lst("z_pl"), inject_pl=4, noise_pl=lst("noise_pl"), pl_image_noise=z[pre + "pl_image_noise"], **extra)


def step_traces(z, pre):
    """{label: {"grad.<p>", "delta.<p>", "gnorm"}} of one golden iteration, plus the EMA movement.  Movements are
    stored as fp16 fractions of the tensor's largest one (`dscale.<p>`)."""
    steps, ema = {}, {}
    for key in z.keys(pre + "step."):
        label, rest = key[len(pre + "step."):].split(".", 1)
        if rest.startswith("dscale."):
            continue
        value = z[key]
        if rest.startswith("delta."):
            value = value.double() * float(z[pre + "step." + label + ".dscale." + rest[len("delta."):]])
        if label == "ema":
            ema[rest[len("delta."):]] = value
        else:
            steps.setdefault(label, {})[rest] = value
    return steps, ema


def split_trace(trace):
    steps, ema = {}, {}
    for key, v in trace.items():
        label, rest = key.split(".", 1)
        if label == "ema":
            ema[rest[len("delta."):]] = v
        else:
            steps.setdefault(label, {})[rest] = v
    return steps, ema


def test_train_iteration(golden):
    """Three iterations (1, 16, and 32 with the late-training branches: wrongly ordered reals, CutMix augmentation +
    consistency, top-k): losses, R1, path length, and EVERY optimiser step whole -- pre-clip gradients of all
    parameters, the global norm, the parameter movement where the gradient is above rounding noise -- plus the EMA
    movement from an EMA copy that starts away from the generator."""
    import copy
    from tools.gen_golden import TINY_D, TINY_G
    z = golden("train_step")
    g, d = om.Generator(TINY_G), om.Discriminator(TINY_D, no_rfp=True)
    g.load_state_dict(z.state_dict("train.G0.")); d.load_state_dict(z.state_dict("train.D0."))
    dead0 = g.main_convolutions_2[3].modulated_convolution.weight.detach().clone()
    g_ema = copy.deepcopy(g)
    g_ema.load_state_dict(z.state_dict("train.Gema0."))
    og, od = ot.make_optimizers(g, d)
    pl = ot.PathLength()
    top_k = ot.TopK(0, 1)                        # as resumed training sets it (model_wrapper.py:121-123): v = 0.5
    for step, (iteration, late) in enumerate(GOLDEN_ITERATIONS):
        real, draws = load_train_draws(z, step)
        trace = {}
        log = ot.train_iteration(g, d, g_ema, og, od, pl, real, iteration, draws, trace=trace,
                                 resume_training=late, top_k=top_k if late else None)
        pre = f"train.it{step}."
        for key in z.keys(pre + "log."):
            want = float(z[key])
            assert abs(log[key[len(pre + "log."):]] - want) <= 2e-4 * abs(want), key
        want_steps, want_ema = step_traces(z, pre)
        got_steps, got_ema = split_trace(trace)
        assert list(got_steps) == STEP_LABELS[iteration] and sorted(want_steps) == sorted(got_steps)
        for label, want in want_steps.items():
            st = check_step_trace(got_steps[label], want, tol_grad=5e-4, tol_norm=1e-4, tol_delta=2e-3)
            assert st["compared"] > 0.2 * st["total"], (label, st)
        for n, want in want_ema.items():
            assert rel_err(got_ema[n], want) < 2e-3, n
        gp, dp, ep = dict(g.named_parameters()), dict(d.named_parameters()), dict(g_ema.named_parameters())
        for key in z.keys(pre + "G."):
            assert rel_err(gp[key[len(pre + "G."):]], z[key]) < 1e-4, key
        for key in z.keys(pre + "Gema."):
            assert rel_err(ep[key[len(pre + "Gema."):]], z[key]) < 1e-4, key
        for key in z.keys(pre + "D."):
            assert rel_err(dp[key[len(pre + "D."):]], z[key]) < 1e-4, key
    assert rel_err(pl.mean_path_length, z["train.it2.mean_path_length"]) < 1e-4
    assert torch.equal(g.main_convolutions_2[3].modulated_convolution.weight, dead0)   # dead branch untouched


def test_step_trace_check_bites(golden):
    """The comparison itself must fail for the errors it exists to catch: a skipped optimiser step, a step applied
    twice, a clip with the wrong threshold (gradients fine, deltas not), and a skipped / doubled EMA update."""
    z = golden("train_step")
    steps, ema = step_traces(z, "train.it1.")
    want = steps["r1"]                                   # the R1 step: its gradient norm is far above 5, the clip acts
    assert float(want["gnorm"]) > 5.0
    same = {k: v.clone() for k, v in want.items()}
    check_step_trace(same, want, 5e-4, 1e-4, 2e-3)
    for name, factor in (("skipped", 0.0), ("doubled", 2.0)):
        broken = {k: (v * factor if k.startswith("delta.") else v.clone()) for k, v in want.items()}
        with pytest.raises(AssertionError):
            check_step_trace(broken, want, 5e-4, 1e-4, 2e-3)
    # Adam's second/third step depends on the clipped gradient's size through v: a missing clip changes the deltas
    unclipped = {k: v.clone() for k, v in want.items()}
    scale = float(want["gnorm"]) / 5.0
    for k in unclipped:
        if k.startswith("delta."):
            unclipped[k] = unclipped[k] * (1 + 0.05 * min(scale, 2.0))
    with pytest.raises(AssertionError):
        check_step_trace(unclipped, want, 5e-4, 1e-4, 2e-3)
    some = next(iter(ema))
    assert rel_err(ema[some] * 0.0, ema[some]) > 0.5 and rel_err(ema[some] * 2.0, ema[some]) > 0.5


def test_cut_mix_maps_and_wrong_order_draws_are_pinned(golden):
    """The RNG-consuming helpers: with the seeds stored beside the fixtures the oracle's binary CutMix map
    (u_net_2d_discriminator.py:425-448) and its 'permutation' (misc.py:202-213) reproduce what the reference's own
    functions drew when the fixtures were generated."""
    import random
    import numpy as np
    z = golden("train_step")
    seeds = [int(v) for v in z["train.it2.cut_mix_seeds"]]
    for seed, key in zip(seeds, ("cut_mix_map_aug", "cut_mix_map_reg")):
        torch.manual_seed(seed); random.seed(seed)
        assert torch.equal(ot.binary_cut_mix_map(32, 32), z["train.it2." + key])
    np.random.seed(302)
    assert torch.equal(ot.random_permutation(3), z["train.it2.wrong_order_perm"])
    # the identity draw is replaced by the reversal
    class _Fixed:
        @staticmethod
        def choice(r, size):
            return np.arange(size)
    orig, np.random.choice = np.random.choice, _Fixed.choice
    try:
        assert ot.random_permutation(4).tolist() == [3, 2, 1, 0]
    finally:
        np.random.choice = orig


def test_top_k_schedule():
    """loss.py:413-428: v = 1 up to the start mark, 0.5 from the final mark on, linear in between; k = max(1, int(B v))."""
    tk = ot.TopK(starting_iteration=2, final_iteration=6)
    vs = [tk.calc_v() for _ in range(8)]
    assert vs == [1.0, 1.0, 0.875, 0.75, 0.625, 0.5, 0.5, 0.5]
    tk = ot.TopK(0, 1)
    vals, idx = tk(torch.tensor([[0.3], [-1.0], [2.0], [0.1], [0.2]]))
    assert idx.tolist() == [2, 0] and vals.tolist() == [2.0, 0.30000001192092896]
    assert ot.TopK(0, 1)(torch.tensor([[1.0]]))[1].tolist() == [0]          # k never drops below 1


@pytest.mark.parametrize("case", ["wide", "few_samples", "shifted"])
def test_frechet_distance(golden, case):
    """oracle.metrics.frechet_distance against values computed by the reference's own FID._calc_fid / FVD._calc_fvd
    (validation_metrics.py:192-220, 401-429) on stored feature matrices (scipy's sqrtm is threaded: 1e-8 run to run)."""
    from oracle import metrics as omet
    z = golden("metrics")
    got = omet.frechet_distance(z[f"frechet.{case}.real"].numpy(), z[f"frechet.{case}.fake"].numpy())
    want = float(z[f"frechet.{case}.value"])
    assert abs(got - want) <= 1e-6 * abs(want), (got, want)


def test_inception_score_closed_forms():
    """The inception-score arithmetic (validation_metrics.py:126-140) has no callable in the reference besides IS.__call__
    (which needs the Inception weights): checked against what the formula must give -- identical predictions: 1; confident,
    balanced predictions over K classes: K; in between, the numpy statement of exp(E KL)."""
    import numpy as np
    from oracle import metrics as omet
    k, n = 7, 70
    assert abs(omet.inception_score(np.full((n, k), 1.0 / k)) - 1.0) < 1e-12
    eps = 1e-9
    onehot = np.full((n, k), eps / (k - 1))
    onehot[np.arange(n), np.arange(n) % k] = 1.0 - eps
    assert abs(omet.inception_score(onehot) - k) < 1e-5
    rng = np.random.default_rng(0)
    p = rng.dirichlet(np.ones(k), size=n)
    p_y = p.mean(0)
    want = np.exp(np.mean([(row * (np.log(row) - np.log(p_y))).sum() for row in p]))
    assert abs(omet.inception_score(p) - want) < 1e-12 * want
