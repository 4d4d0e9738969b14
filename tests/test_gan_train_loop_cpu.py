"""The generator half of the reference's GAN loop (gan_train.py:87-134) run, call for call, against the
nmt_gan_amd facade on the CPU: every attribute and keyword the driver uses must exist.  The HIP entry
points are replaced by recording stand-ins (no GPU here), so this checks the SURFACE - names, argument
meaning, shapes, dtypes, call order - not arithmetic; the GPU suite checks the arithmetic."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from helpers import Cfg, disc_weights, facade_config, tiny_cfgs
from oracle import nmtgan_oracle as O

T, B, V = 10, 4, 512


class FakeLib(object):
    """Stand-ins for the entry points the loop reaches; they write plausible outputs into the (host) tensors
    whose addresses the facade passes, and log the call order."""

    def __init__(self):
        self.calls = []

    def nmtgan_last_error(self):
        return b""

    def nmtgan_generate_workspace_bytes(self, g, B_, Ls, T_):
        return 256

    def nmtgan_generate_greedy(self, g, x, pad, B_, Ls, T_, y, ws, n, st):
        self.calls.append("generate_greedy")
        out = np.ctypeslib.as_array((C.c_int32 * (B_ * T_)).from_address(y)).reshape(B_, T_)
        out[:] = 7 + np.arange(B_ * T_).reshape(B_, T_) % 50
        out[0, 6] = 3                    # </S> inside row 0
        out[1, 0] = 3                    # empty sample
        return 0

    def nmtgan_rollout_reward_workspace_bytes(self, g, d, plan):
        return 256

    def nmtgan_rollout_reward(self, g, d, plan, ws, n, st):
        self.calls.append("rollout_reward")
        p = plan._obj
        assert p.T == T and p.rollnum == 3 and p.use_bias == 1 and abs(p.bias_num - 0.6) < 1e-12
        mask = np.ctypeslib.as_array((C.c_float * (p.B * p.T)).from_address(p.mask)).reshape(p.B, p.T)
        rew = np.ctypeslib.as_array((C.c_float * (p.B * p.T)).from_address(p.rewards)).reshape(p.B, p.T)
        rew[:] = (0.25 - 0.6) * mask
        self.last_uniform0 = float(np.ctypeslib.as_array((C.c_float * 1).from_address(p.uniforms))[0])
        return 0

    def nmtgan_pg_workspace_bytes(self, g, B_, Ls, T_):
        return 256

    def nmtgan_pg_grads(self, g, gr, x, pad, y, dec_in, reward, B_, Ls, T_, loss, p_target, ws, n, st):
        self.calls.append("pg_grads")
        np.ctypeslib.as_array((C.c_float * 1).from_address(loss))[0] = -0.125
        return 0

    def nmtgan_rmsprop_update(self, *a):
        self.calls.append("rmsprop_update")
        return 0

    def nmtgan_clip_values(self, *a):       # the default (fp16) tier refreshes its scoring copy from the fp32 master
        self.calls.append("clip_values")
        return 0


class StubDataUtil(object):
    """What gan_train.py reads from utils.DataUtil inside the loop: the target dictionary and the batch iterator."""

    def __init__(self):
        self.dst2idx = {"<PAD>": 0, "<UNK>": 1, "<S>": 2, "</S>": 3}

    def get_training_batches(self, set_train_src_path=None, set_train_dst_path=None, set_batch_size=None,
                             set_max_length=None):
        rng = np.random.default_rng(0)
        while True:
            x = rng.integers(4, V, (set_batch_size, set_max_length - 1)).astype(np.int32)
            x = np.concatenate([x, np.full((set_batch_size, 1), 3, np.int32)], axis=1)
            y = [np.concatenate([rng.integers(4, V, n), [3]]).astype(np.int32) for n in (3, 9, 5, 1)[:set_batch_size]]
            yield x, y


@pytest.fixture
def stubbed(monkeypatch, tmp_path):
    import nmt_gan_amd as N
    from nmt_gan_amd import _lib, cnn_discriminator, model
    fake = FakeLib()
    monkeypatch.setattr(model, "lib", fake)
    monkeypatch.setattr(cnn_discriminator, "lib", fake)
    monkeypatch.setattr(_lib, "stream_ptr", lambda: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    gcfg, dcfg = tiny_cfgs(T=T)
    config = N.AttrDict(dict(facade_config(gcfg)))
    config.update(rollnum=3, bias_num=0.6, gan_iter_num=1, gan_gen_iter_num=2,
                  generator=N.AttrDict(max_length=T, devices="0", optimizer="rmsprop", batch_size=B, src_path="s",
                                       dst_path="t", modelFile=str(tmp_path / "gen" / "model"), teacher_forcing=True,
                                       learning_rate=1e-5),
                  discriminator=N.AttrDict(dis_max_len=T, dis_batch_size=B, dis_dim_word=dcfg.dim_word,
                                           dis_saveto=str(tmp_path / "disc" / "d")))
    os.makedirs(tmp_path / "gen")
    os.makedirs(tmp_path / "disc")
    return N, fake, config, gcfg, dcfg


def test_generator_loop_of_gan_train_runs_against_the_facade(stubbed):
    N, fake, config, gcfg, dcfg = stubbed
    from nmt_gan_amd import Model, DisCNN
    from nmt_gan_amd import deal_generated_samples, deal_generated_samples_to_maxlen, extend_sentence_to_maxlen
    du = StubDataUtil()
    sess, default_graph = object(), object()
    # gan_train.py:36-45
    generator = Model(config=config, graph=default_graph, sess=sess, device="cpu")
    generator.overlap = False                      # no HIP streams on the CPU
    generator.build_train_model()
    generator.build_generate(max_len=config.generator.max_length, generate_devices=config.generator.devices,
                             optimizer=config.generator.optimizer)
    generator.build_rollout_generate(max_len=config.generator.max_length, roll_generate_devices=config.generator.devices)
    generator.init_and_restore(weights=O.init_generator_weights(gcfg, 0))
    # gan_train.py:47-78 (the data-path keywords are accepted and ignored)
    discriminator = DisCNN(sess=sess, max_len=config.discriminator.dis_max_len, num_classes=2,
                           vocab_size=config.dst_vocab_size, vocab_size_s=config.src_vocab_size,
                           batch_size=config.discriminator.dis_batch_size, dim_word=config.discriminator.dis_dim_word,
                           filter_sizes=dcfg.filter_sizes, num_filters=dcfg.num_filters, source_dict=None, target_dict=None,
                           gpu_device="gpu-0", positive_data=None, negative_data=None, source_data=None,
                           dev_positive_data=None, dev_negative_data=None, dev_source_data=None, max_epoches=1, dispFreq=1,
                           saveFreq=1, saveto=config.discriminator.dis_saveto, reload=False, clip_c=1.0, optimizer="rmsprop",
                           reshuffle=False, scope="discnn", device="cpu")
    discriminator.load_weights(disc_weights(dcfg, 1))
    batch_iter = du.get_training_batches(set_train_src_path=config.generator.src_path,
                                         set_train_dst_path=config.generator.dst_path,
                                         set_batch_size=config.generator.batch_size,
                                         set_max_length=config.generator.max_length)
    seen_u = []
    # gan_train.py:87-134
    for epoch in range(1, config.gan_iter_num + 1):
        for gen_iter in range(config.gan_gen_iter_num):
            batch = next(batch_iter)
            x, y_ground = batch[0], batch[1]
            y_sample = generator.generate_step(x)
            y_sample_dealed, y_sample_mask = deal_generated_samples(y_sample, du.dst2idx)
            x_to_maxlen = extend_sentence_to_maxlen(x, config.generator.max_length)
            rewards = generator.get_reward(x=x, x_to_maxlen=x_to_maxlen, y_sample=y_sample_dealed,
                                           y_sample_mask=y_sample_mask, rollnum=config.rollnum, disc=discriminator,
                                           max_len=config.discriminator.dis_max_len, bias_num=config.bias_num, data_util=du)
            loss = generator.generate_step_and_update(x, y_sample_dealed, rewards)
            generator.saver.save(generator.sess, config.generator.modelFile)
            seen_u.append(fake.last_uniform0)
            assert rewards.shape == (B, T) and rewards.dtype == np.float32 and loss == pytest.approx(-0.125)
            assert y_sample.shape == (B, T) and y_sample.dtype == np.int32
            assert y_sample_mask[0].sum() == 6 and y_sample_mask[1].sum() == 0 and (y_sample_dealed[0, 6:] == 0).all()
            assert np.array_equal(rewards != 0, y_sample_mask != 0)
            if config.generator.teacher_forcing:
                y_ground, y_ground_mask = deal_generated_samples_to_maxlen(y_sample=y_ground, dicts=du.dst2idx,
                                                                           maxlen=config.discriminator.dis_max_len)
                rewards_ground = np.ones_like(y_ground)
                rewards_ground = rewards_ground * y_ground_mask
                loss = generator.generate_step_and_update(x, y_ground, rewards_ground)
                assert y_ground.shape == (B, T) and y_ground_mask.sum(axis=1).tolist() == [3, 9, 5, 1]
        generator.saver.save(generator.sess, config.generator.modelFile)
    # gan_train.py:161
    discriminator.saver.save(discriminator.sess, discriminator.saveto)
    per_iter = ["generate_greedy", "rollout_reward", "pg_grads", "rmsprop_update", "pg_grads", "rmsprop_update"]
    # (the default split-precision tier rebuilds its [hi | hi | lo] scoring copy from the fp32 master after every
    # update with torch ops, no library call; the fp16 tier refreshes its fp16 copy through clip_values)
    assert fake.calls == per_iter * 2
    assert generator.precision == "fp16x3" and discriminator.precision == "fp16x3"
    # fresh noise per get_reward call (tf.multinomial is unseeded, model.py:239), reproducible per Model
    assert seen_u[0] != seen_u[1]
    # the bundles the two savers wrote restore into fresh facades
    g2 = Model(config=config, device="cpu")
    g2.saver.restore(None, config.generator.modelFile)
    for k, v in generator.get_weights().items():
        assert np.array_equal(g2.get_weights()[k], v), k
    d2 = DisCNN(None, T, 2, V, B, dcfg.dim_word, dcfg.filter_sizes, dcfg.num_filters, vocab_size_s=V, reload=True,
                saveto=discriminator.saveto, device="cpu")
    for k, v in discriminator.get_weights().items():
        assert np.array_equal(d2.get_weights()[k], v), k


def test_default_seed_advances_and_is_rank_independent(stubbed):
    """Two Models with the same reward_seed_base (= two ranks of a job) draw the same uniforms call after call,
    and every call draws new ones; seed= and uniforms= still override."""
    N, fake, config, gcfg, dcfg = stubbed
    models = []
    for _ in range(2):
        m = N.Model(config=config, device="cpu")
        m.overlap = False
        m.build_rollout_generate(T)
        m.init_and_restore(weights=O.init_generator_weights(gcfg, 0))
        models.append(m)
    disc = N.DisCNN(None, T, 2, V, B, dcfg.dim_word, dcfg.filter_sizes, dcfg.num_filters, vocab_size_s=V, device="cpu")
    disc.load_weights(disc_weights(dcfg, 1))
    x, y = O.synthetic_batch(B, T, V, V)
    yd, mask = N.deal_generated_samples(y, None)
    xp = N.extend_sentence_to_maxlen(x, T)
    seq = [[], []]
    for i, m in enumerate(models):
        for _ in range(3):
            m.get_reward(x, xp, yd, mask, 3, disc, max_len=T, bias_num=0.6)
            seq[i].append(fake.last_uniform0)
    assert seq[0] == seq[1] and len(set(seq[0])) == 3
    models[0].get_reward(x, xp, yd, mask, 3, disc, max_len=T, bias_num=0.6, seed=5)
    a = fake.last_uniform0
    models[1].get_reward(x, xp, yd, mask, 3, disc, max_len=T, bias_num=0.6, seed=5)
    assert fake.last_uniform0 == a
    u = np.full((3, T - 1, B, T), 0.375, np.float32)
    models[0].get_reward(x, xp, yd, mask, 3, disc, max_len=T, bias_num=0.6, uniforms=u)
    assert fake.last_uniform0 == 0.375


def test_reference_named_eos_helpers_follow_the_reference():
    import nmt_gan_amd as N
    ragged = [np.array([5, 6, 3, 7]), np.array([4, 4]), np.array([3])]
    y, m = N.deal_generated_samples(ragged, {"</S>": 3})
    assert y.tolist() == [[5, 6, 0, 0], [4, 4, 0, 0], [0, 0, 0, 0]] and m.sum(axis=1).tolist() == [2, 2, 0]
    y, m = N.deal_generated_samples_to_maxlen(ragged, {"</S>": 3}, 6)
    assert y.shape == (3, 6) and y[0].tolist() == [5, 6, 0, 0, 0, 0] and m.dtype == np.float32 and y.dtype == np.int32
    # a dictionary without '</S>' cuts nothing (dicts.get -> None never matches, share_function.py:272, 284-285)
    y, m = N.deal_generated_samples(np.array([[5, 3, 6]], np.int32), {"a": 1})
    assert y.tolist() == [[5, 3, 6]] and m.tolist() == [[1, 1, 1]]
    with pytest.raises(ValueError):
        N.deal_generated_samples_to_maxlen([np.arange(4, 12)], {"</S>": 3}, 5)
