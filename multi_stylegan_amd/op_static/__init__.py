"""Drop-in for the reference's ``multi_stylegan.op_static`` (op_static/__init__.py:1-2), backed by libmsg_hip.so."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu, fused_bias_noise_leaky_relu, gamma_merge, scaled_add, scaled_add_fork
from .upfirdn2d import upfirdn2d, blur_bias_act
from .softmax import softmax_rows
from .attention import non_local_attention
from .maxpool import max_pool2x2
from . import rgb_skip

__all__ = ["FusedLeakyReLU", "fused_leaky_relu", "fused_bias_noise_leaky_relu", "gamma_merge", "scaled_add", "scaled_add_fork", "upfirdn2d", "blur_bias_act", "softmax_rows", "non_local_attention", "max_pool2x2"]
