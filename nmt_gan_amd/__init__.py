"""nmt_gan_amd - MI355X-native rollout-reward path of NMT_GAN (Model.get_reward and its callers'
boundary), as hand-written gfx950 HIP kernels behind a C ABI (include/nmtgan.h).

Importing the package loads libnmtgan_hip.so and raises if it is missing: there is no fallback.
"""
from ._lib import lib, LIB_PATH, NmtGanError          # noqa: F401  (loads the HIP library)
from .utils import AttrDict                            # noqa: F401
from .share_function import (deal_generated_samples, deal_generated_samples_to_maxlen,   # noqa: F401
                             extend_sentence_to_maxlen, FlushFile, build_rollout_plan, split_bounds,
                             timing_signal, prepare_gan_dis_data)
from .model import Model                               # noqa: F401
from .cnn_discriminator import DisCNN                  # noqa: F401

__all__ = ["Model", "DisCNN", "AttrDict", "deal_generated_samples", "deal_generated_samples_to_maxlen",
           "extend_sentence_to_maxlen", "FlushFile",
           "build_rollout_plan", "split_bounds", "timing_signal", "prepare_gan_dis_data", "lib", "LIB_PATH", "NmtGanError"]
