"""The loop body of gan_train.py:87-161 end to end on the MI355X path (tiny model): generate_step ->
deal_generated_samples -> get_reward -> generate_step_and_update -> discriminator train_step on generated + reference
sentences -> the next iteration's get_reward sees both updated networks."""
import numpy as np
import pytest

from helpers import Cfg, disc_weights, make_gpu_pair, tiny_cfgs
from oracle import nmtgan_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["fp32", "fp16", "bf16", "fp16x3"])
def test_two_gan_iterations(cuda, tmp_path, precision):
    import nmt_gan_amd as N
    T, B, R = 10, 4, 2
    gcfg, dcfg = tiny_cfgs(T=T)
    Wg, Wd = O.init_generator_weights(gcfg, 0), disc_weights(dcfg, 1)
    gen, disc = make_gpu_pair(gcfg, dcfg, Wg, Wd, precision, T)
    gen.config["generator"] = Cfg(learning_rate=1e-3)
    x, y_ground = O.synthetic_batch(B, T, gcfg.src_vocab_size, gcfg.dst_vocab_size, seed=4, short_src_rows=(1,))
    rewards_seen = []
    for it in range(2):
        y_sample = gen.generate_step(x)
        y, mask = N.deal_generated_samples(y_sample)
        xp = N.extend_sentence_to_maxlen(x, T)
        u = np.random.default_rng(it).random((R, T - 1, B, T), dtype=np.float32)
        rewards = gen.get_reward(x, xp, y, mask, R, disc, max_len=T, bias_num=0.6, uniforms=u)
        assert rewards.shape == (B, T) and np.isfinite(rewards).all()
        # the facade's rewards are the oracle's for the CURRENT weights of both networks (fp32 tier)
        if precision == "fp32":
            ref = O.get_reward(O.to_torch(gen.get_weights()), gcfg, O.to_torch(disc.get_weights()), dcfg, x, xp, y, mask, R, T,
                               0.6, u, mode="incremental")
            assert np.abs(rewards - ref).max() < 2e-5, it
        rewards_seen.append(rewards)
        loss = gen.generate_step_and_update(x, y, rewards * np.float32(20.0))
        assert np.isfinite(loss)
        yg, _ = N.deal_generated_samples(y_ground)
        lab = np.zeros((2, 2 * B), np.float32)
        lab[0, :B], lab[1, B:] = 1.0, 1.0
        d_loss, d_acc = disc.train_step(np.concatenate([y, yg]).T, lab, np.concatenate([xp, xp]).T)
        assert np.isfinite(d_loss) and 0.0 <= d_acc <= 1.0
    gen.save(str(tmp_path / "g" / "model"))
    disc.save(str(tmp_path / "d" / "disc"))
    gen2, disc2 = make_gpu_pair(gcfg, dcfg, Wg, Wd, precision, T)
    gen2.init_and_restore(modelFile=str(tmp_path / "g" / "model"))
    disc2.restore(str(tmp_path / "d" / "disc"))
    u = np.random.default_rng(9).random((R, T - 1, B, T), dtype=np.float32)
    y, mask = N.deal_generated_samples(gen.generate_step(x))
    assert np.array_equal(gen2.generate_step(x), gen.generate_step(x))
    xp = N.extend_sentence_to_maxlen(x, T)
    a = gen.get_reward(x, xp, y, mask, R, disc, max_len=T, bias_num=0.6, uniforms=u)
    b = gen2.get_reward(x, xp, y, mask, R, disc2, max_len=T, bias_num=0.6, uniforms=u)
    assert np.abs(a - b).max() < 1e-5
