"""Shared test helpers: tiny configs, oracle weights, GPU facades loaded with them."""
import numpy as np

from oracle import nmtgan_oracle as O


class Cfg(dict):
    __getattr__ = dict.get


def tiny_cfgs(d=64, blocks=2, heads=4, Vs=512, Vt=512, T=12, E=64, shared=False):
    gcfg = O.GenConfig(hidden_units=d, num_blocks=blocks, num_heads=heads, src_vocab_size=Vs,
                       dst_vocab_size=Vt, scale_embedding=True, shared_embedding=shared)
    dcfg = O.DisConfig(max_len=T, dim_word=E, vocab_size=Vt, vocab_size_s=Vs)
    return gcfg, dcfg


def facade_config(gcfg):
    return Cfg(hidden_units=gcfg.hidden_units, num_blocks=gcfg.num_blocks, num_heads=gcfg.num_heads,
               src_vocab_size=gcfg.src_vocab_size, dst_vocab_size=gcfg.dst_vocab_size,
               scale_embedding=gcfg.scale_embedding, residual_dropout_rate=0.0,
               train=Cfg(shared_embedding=gcfg.shared_embedding))


def disc_weights(dcfg, seed=1, head_scale=0.05):
    """Reference initialisers, with the 2-way head shrunk so P(real) is not saturated at 0/1
    (a saturated head would make the reward comparison vacuous)."""
    Wd = O.init_disc_weights(dcfg, seed)
    Wd["discnn/model_self/W"] = (Wd["discnn/model_self/W"] * head_scale).astype(np.float32)
    return Wd


def make_gpu_pair(gcfg, dcfg, Wg, Wd, precision, max_len):
    import nmt_gan_amd as N
    m = N.Model(facade_config(gcfg), precision=precision)
    m.build_train_model()
    m.build_generate(max_len)
    m.build_rollout_generate(max_len)
    m.init_and_restore(weights=Wg)
    disc = N.DisCNN(None, max_len, 2, dcfg.vocab_size, 8, dcfg.dim_word, dcfg.filter_sizes, dcfg.num_filters,
                    vocab_size_s=dcfg.vocab_size_s, precision=precision)
    disc.load_weights(Wd)
    return m, disc


def rel_err(a, b):
    """tensor-scale relative error ||a-b||_inf / ||b||_inf."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
