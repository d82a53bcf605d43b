"""TEST INFRASTRUCTURE -- CPU restatement (numpy / scipy) of the sample-quality STATISTICS of the reference's
multi_stylegan/validation_metrics.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import oracle/.

What is restated: the arithmetic on features -- not the feature networks (torchvision's pretrained Inception-v3 and the I3D
video network, whose weights are missing blobs of the reference: .MISSING_LARGE_BLOBS).

* ``frechet_distance``  -- FID._calc_fid (validation_metrics.py:192-220) and FVD._calc_fvd (:401-429): the two are the same
  statement.  PINNED: tools/gen_golden.py calls the reference's own static methods and stores features + result in
  tests/golden/metrics.npz; tests/test_oracle_golden.py holds this function to them.
* ``inception_score``   -- IS.__call__ (:126-140): ``exp(mean_i sum_c p_ic log(p_ic / mean_j p_jc))`` on softmax outputs.  The
  reference has no callable for it apart from ``__call__`` itself (which needs the Inception weights): PARITY UNPINNED,
  checked against the closed form instead (uniform predictions -> 1, one-hot over K balanced classes -> K).
* ``select_frames``     -- the frame choice of :93-102 / :246-256: channel c of a random time step, replicated to three
  colour planes, one ``torch.randint`` draw per channel and batch.
"""
import numpy as np
from scipy.linalg import sqrtm


def frechet_distance(real_activations: np.ndarray, fake_activations: np.ndarray) -> float:
    """validation_metrics.py:192-220 / :401-429 -- ||mu_r - mu_f||^2 + tr(C_r) + tr(C_f) - 2 tr(sqrtm(C_r C_f)), covariances
    with numpy's N - 1 normalisation, the imaginary part of the matrix square root dropped."""
    real_mu, fake_mu = np.mean(real_activations, axis=0), np.mean(fake_activations, axis=0)
    real_cov, fake_cov = np.cov(real_activations, rowvar=False), np.cov(fake_activations, rowvar=False)
    diff = real_mu - fake_mu
    cov_mean, _ = sqrtm(real_cov @ fake_cov, disp=False)
    if np.iscomplexobj(cov_mean):
        cov_mean = cov_mean.real
    return float(diff @ diff + np.trace(real_cov) + np.trace(fake_cov) - 2 * np.trace(cov_mean))


def inception_score(predictions: np.ndarray) -> float:
    """validation_metrics.py:126-140 -- predictions [samples, classes] are softmax outputs."""
    p_y = predictions.mean(axis=0, keepdims=True)
    kl = np.sum(predictions * np.log(predictions / p_y), axis=-1)
    return float(np.exp(kl.mean()))


def select_frames(images, channel: int, t: int):
    """validation_metrics.py:93-94 -- ``images[:, c, randint].unsqueeze(1).repeat_interleave(3, 1)`` for a drawn time step t:
    [B, C, T, H, W] -> [B, 3, 1, H, W] (the singleton is the one-element index tensor's axis; callers take ``[:, :, 0]``)."""
    return images[:, channel, t:t + 1][:, None].repeat(3, axis=1) if isinstance(images, np.ndarray) else \
        images[:, channel, t:t + 1].unsqueeze(1).repeat_interleave(3, dim=1)
