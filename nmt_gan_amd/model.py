"""Model facade: the generator surface gan_train.py / generate_samples.py call in the reference
(model.py:18, 203-288, 291-346, 425-435), backed by libnmtgan_hip.so on one MI355X per process.

Built: get_reward (the hot path), generate_step / generate_and_save (greedy sampler), generate_step_and_update
(policy-gradient update), init_and_restore / saver.save on TF V2 checkpoint bundles.  The MLE training graph,
beam search and get_reward_Obinforced are not built: build_train_model only records is_training,
get_reward_Obinforced raises NotImplementedError.

Differences from the reference that a caller can see:
  * numpy in / numpy out is unchanged; `uniforms=` / `seed=` are keyword-only extensions that
    replace TF's unseeded Philox stream (model.py:239) with injected uniforms u[R, T-1, B, T].
  * residual dropout, which the reference leaves ON during rollouts, is
    neutralised: the path is deterministic given the uniforms.
  * a call without `uniforms=` / `seed=` draws fresh noise: its seed is `reward_seed_base` plus the number of
    get_reward calls this Model has served (deterministic, and identical on every rank of a sharded job).
  * with torch.distributed initialised (one process per GPU, RCCL), get_reward shards the sentences
    across ranks with the reference's split_tensor boundaries and all-gathers the reward rows; with fewer
    sentences than ranks it shards (sentence, roll) units instead and the per-rollout scores are gathered and
    summed in roll order on every rank.
"""
from __future__ import annotations

import codecs
import ctypes as C
import logging
import math
import weakref
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr
from .share_function import (build_rollout_plan, deal_generated_samples, roll_shard_pieces, shift_right,
                             split_bounds, timing_signal)

INT_TYPE = np.int32
FLOAT_TYPE = np.float32


def _cfg_get(cfg, path, default=None):
    cur = cfg
    for k in path.split("."):
        if cur is None:
            return default
        cur = cur.get(k) if isinstance(cur, dict) else getattr(cur, k, None)
    return default if cur is None else cur


class _Saver(object):
    """Stand-in for the `saver` attribute (a tf.train.Saver) the reference's drivers use:
    `generator.saver.save(generator.sess, path)` (gan_train.py:118, 134), `discriminator.saver.save(
    discriminator.sess, discriminator.saveto)` (gan_train.py:161), `saver.restore(sess, path)` (model.py:432).
    `sess` is accepted and ignored; the files are TF V2 checkpoint bundles.
    The owner is held weakly: a model and its saver form no reference cycle, so dropping the last reference to a model
    returns its device memory (tens of GiB of workspace at benchmark size) at once instead of at the next cyclic
    garbage collection."""

    def __init__(self, owner, save_name, restore_name):
        self._owner = weakref.ref(owner)
        self._save_name, self._restore_name = save_name, restore_name

    def _call(self, name, path):
        owner = self._owner()
        if owner is None:
            raise RuntimeError("saver: the model it belongs to no longer exists")
        getattr(owner, name)(path)

    def save(self, sess, save_path, global_step=None, **_ignored):
        path = str(save_path) if global_step is None else "%s-%d" % (save_path, int(global_step))
        self._call(self._save_name, path)
        return path

    def restore(self, sess, save_path):
        self._call(self._restore_name, str(save_path))


class Model(object):
    def __init__(self, config, graph=None, sess=None, precision="fp16x3", device="cuda:0", max_pos=256):
        # precision: 'fp16x3' (default) is the fastest tier that meets the whole stated tolerance at V = 32000 (logits
        # and rewards within 1e-3 of fp32, >= 99.5 % of the sampled tokens equal); 'fp16' is the 16-bit throughput tier
        # (2.2x faster; logits within 1e-3, about 9 % of the draws at V = 32000 land in another CDF bucket);
        # 'bf16'; 'fp32' = exact fp32 MFMA, the parity tier
        self.graph = graph
        self.sess = sess
        self.config = config
        self._logger = logging.getLogger('model')
        self.precision = precision
        self.device = torch.device(device)
        # 'fp16x3': fp32 storage and attention as in 'fp32', every GEMM as one fp16 MFMA product over split operands
        self.split = precision == _lib.SPLIT_PRECISION
        self._tdtype, self._op = _lib.op_dtype_of("fp32" if self.split else precision)
        self.hidden_units = int(_cfg_get(config, "hidden_units"))
        self.num_blocks = int(_cfg_get(config, "num_blocks"))
        self.num_heads = int(_cfg_get(config, "num_heads"))
        self.src_vocab_size = int(_cfg_get(config, "src_vocab_size"))
        self.dst_vocab_size = int(_cfg_get(config, "dst_vocab_size"))
        self.scale_embedding = bool(_cfg_get(config, "scale_embedding", True))
        self.shared_embedding = bool(_cfg_get(config, "train.shared_embedding", False))
        if self.num_blocks > _lib.MAX_BLOCKS:
            raise ValueError("num_blocks > %d" % _lib.MAX_BLOCKS)
        self.max_pos = int(max_pos)
        self.max_len = None
        self.is_training = False
        self._keep, self._c = [], None
        self._ws = None
        self.logits_chunk_rows = 4096
        self.disc_chunk_rows = 4096    # rows per discriminator pass: one run of equal give_num / 4 at C2 (64 x 16 x 4 rows)
        # two-stream schedule of get_reward (discriminator scoring of finished rollouts overlaps the decode steps)
        self.overlap = True
        self.group_slots = True        # 32 // rollnum prefix groups of a sentence per attention workgroup
        self.conv_bucket_taps = 0      # partial-window reuse in the discriminator: 0 = library default, < 0 off
        self.conv_table = 0            # table form of the rollouts' conv (per-(token, tap) filter responses of the whole
        #                                vocabulary, rebuilt inside every get_reward): 0 = library default (on, except fp32 tier), > 0 on, < 0 off
        self.conv_table_max_bytes = 0  # 0 = library default (96 GiB of workspace for the tables)
        self.aux_gemm_wgs = 0          # 0 = library default
        self.aux_min_rows = 0
        self._aux_stream = None
        self.last_details = None
        self.g_optimizer = 'rmsprop'
        self._W_src, self._train = None, None
        self.reward_seed_base = 0      # seed of the n-th get_reward call without uniforms=/seed= is base + n
        self._reward_calls = 0
        self._gather_bufs = {}
        self.saver = _Saver(self, "save", "_restore_from")

    # ------------------------------------------------------------------ graph-building surface
    def build_train_model(self):
        """Reference: model.py:67-106 (MLE training graph; also flips is_training, which makes the
        reference apply dropout during rollouts - quirk B.1, neutralised here)."""
        self.is_training = True
        rate = float(_cfg_get(self.config, "residual_dropout_rate", 0.0) or 0.0)
        if rate != 0.0:
            self._logger.warning("residual_dropout_rate=%g ignored on the rollout path (deterministic rollouts)", rate)

    def build_generate(self, max_len, generate_devices=None, optimizer='rmsprop'):
        """Reference: model.py:108-201: the greedy sampler (generate_step) and the policy-gradient update
        (generate_step_and_update).  As in the reference only 'adadelta' selects anything but RMSProp: its
        `if adam / if adadelta / else rmsprop` chain (model.py:111-119) overwrites the Adam choice."""
        self.max_len = int(max_len)
        self.g_optimizer = 'adadelta' if optimizer == 'adadelta' else 'rmsprop'

    def build_rollout_generate(self, max_len, roll_generate_devices=None):
        """Reference: model.py:203-266."""
        self.max_len = int(max_len)

    def init_and_restore(self, modelFile=None, weights: Optional[Dict[str, np.ndarray]] = None):
        """Reference: model.py:425-435 (tf.train.Saver restore of the trainable variables from `modelFile`,
        or from the latest checkpoint of config.train.logdir when it is None).  `modelFile` is a TF V2
        checkpoint prefix (read by nmt_gan_amd.tf_checkpoint, no TensorFlow) or a .npz with the same
        variable names; `weights=` takes the dict directly."""
        if weights is None:
            from . import tf_checkpoint
            if modelFile is None:
                logdir = _cfg_get(self.config, "train.logdir")
                modelFile = tf_checkpoint.latest_checkpoint(logdir) if logdir else None
                if modelFile is None:
                    raise ValueError("init_and_restore: no modelFile and no checkpoint under config.train.logdir")
            if str(modelFile).endswith(".npz"):
                with np.load(modelFile, allow_pickle=False) as z:
                    weights = {k: z[k] for k in z.files}
            else:
                weights = tf_checkpoint.LazyCheckpoint(str(modelFile))
        self.load_weights(weights)

    def load_weights(self, W: Dict[str, np.ndarray]):
        dev, td = self.device, self._tdtype
        d = self.hidden_units
        keep = []

        def dv(a, dtype):
            t = torch.as_tensor(np.ascontiguousarray(a)).to(device=dev, dtype=dtype).contiguous()
            keep.append(t)
            return t

        def kernel2d(name):
            k = np.asarray(W[name + "/kernel"], np.float32)
            return k.reshape(k.shape[-2], k.shape[-1])            # conv kernels are [1,1,din,dout]

        def mat(a):                                               # Wt[N][K] in the op dtype, or its [N][3K] split form
            if self.split:
                t = _lib.split3_weights(torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev))
                keep.append(t)
                return t
            return dv(a, td)

        def dense(cd, name):
            cd.w = ptr(mat(kernel2d(name).T))
            cd.b = ptr(dv(np.asarray(W[name + "/bias"], np.float32), torch.float32))

        def lnorm(cl, name):
            cl.g = ptr(dv(np.asarray(W[name + "/layer_norm_scale"], np.float32), torch.float32))
            cl.b = ptr(dv(np.asarray(W[name + "/layer_norm_bias"], np.float32), torch.float32))

        c = _lib.Generator()
        c.op_dtype = self._op
        c.gemm_split = 1 if self.split else 0
        c.d, c.heads, c.n_blocks = d, self.num_heads, self.num_blocks
        c.vocab_src, c.vocab_dst = self.src_vocab_size, self.dst_vocab_size
        c.emb_mult = float(np.float32(d ** 0.5)) if self.scale_embedding else 1.0     # model.py:494, 553
        if self.shared_embedding:
            shared = dv(np.asarray(W["embedding/shared/kernel"], np.float32), torch.float32)
            c.emb_src = c.emb_dst = ptr(shared)
            c.softmax_w = ptr(mat(np.asarray(W["embedding/shared/kernel"], np.float32)))
        else:
            c.emb_src = ptr(dv(np.asarray(W["embedding/src_embedding/kernel"], np.float32), torch.float32))
            c.emb_dst = ptr(dv(np.asarray(W["embedding/dst_embedding/kernel"], np.float32), torch.float32))
            c.softmax_w = ptr(mat(np.asarray(W["embedding/softmax/kernel"], np.float32)))
        c.timing = ptr(dv(timing_signal(self.max_pos, d), torch.float32))
        c.max_pos = self.max_pos
        for i in range(self.num_blocks):
            p = "encoder/block_%d/" % i
            e = c.enc[i]
            dense(e.qkv, p + "encoder_self_attention/qkv_transform_single")
            dense(e.o, p + "encoder_self_attention/output_transform_single")
            dense(e.ffn1, p + "conv_hidden_relu/conv1_single")
            dense(e.ffn2, p + "conv_hidden_relu/conv2_single")
            lnorm(e.ln0, p + "layer_norm")
            lnorm(e.ln1, p + "layer_norm_1")
            p = "decoder/block_%d/" % i
            b = c.dec[i]
            dense(b.qkv, p + "decoder_self_attention/qkv_transform_single")
            dense(b.o, p + "decoder_self_attention/output_transform_single")
            dense(b.cq, p + "decoder_vanilla_attention/q_transform_single")
            dense(b.ckv, p + "decoder_vanilla_attention/kv_transform_single")
            dense(b.co, p + "decoder_vanilla_attention/output_transform_single")
            dense(b.ffn1, p + "conv_hidden_relu/conv1_single")
            dense(b.ffn2, p + "conv_hidden_relu/conv2_single")
            lnorm(b.ln0, p + "layer_norm")
            lnorm(b.ln1, p + "layer_norm_1")
            lnorm(b.ln2, p + "layer_norm_2")
        self._keep, self._c = keep, c
        self._W_src, self._train = W, None
        return self

    @property
    def c_struct(self):
        if self._c is None:
            raise RuntimeError("Model: init_and_restore()/load_weights() has not been called")
        return self._c

    def _workspace(self, need):
        if need == 0:
            raise _lib.NmtGanError("workspace query failed: " + lib.nmtgan_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _dev_i32(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)

    # ------------------------------------------------------------------ greedy generation (row f-1)
    def generate_step(self, sentence_x):
        """Reference: model.py:268-271 -> greedy while_loop model.py:140-176.  int32 [B, max_len]."""
        if self.max_len is None:
            raise RuntimeError("call build_generate(max_len, ...) first")
        x = np.ascontiguousarray(np.asarray(sentence_x), dtype=np.int32)
        B, Ls = x.shape
        T = self.max_len
        xd = self._dev_i32(x)
        pad = (xd == 0).to(torch.uint8).contiguous()
        y = torch.empty((B, T), dtype=torch.int32, device=self.device)
        g = self.c_struct
        ws = self._workspace(lib.nmtgan_generate_workspace_bytes(C.byref(g), B, Ls, T))
        check(lib.nmtgan_generate_greedy(C.byref(g), ptr(xd), ptr(pad), B, Ls, T, ptr(y), ptr(ws), ws.numel(),
                                         _lib.stream_ptr()), "nmtgan_generate_greedy")
        torch.cuda.synchronize(self.device)
        return y.cpu().numpy()

    def generate_and_save(self, data_util, infile, generate_batch, outfile):
        """Reference: model.py:278-288."""
        with codecs.open(outfile, 'w', 'utf-8') as out:
            for batch in data_util.get_test_batches(infile, generate_batch):
                out_generate = self.generate_step(batch)
                dealed, _ = deal_generated_samples(out_generate, data_util.dst2idx)
                for y_str in data_util.indices_to_words_del_pad(dealed, 'dst'):
                    out.write(y_str + '\n')

    # ------------------------------------------------------------------ policy-gradient update (row f-3)
    def _param_specs(self):
        """(TF variable name, kind, path into the nmtgan_generator struct) of every trainable variable
       .  kind: mat = dense kernel packed Wt[N][K] and read in op dtype, vec = fp32 vector,
        tab = fp32 embedding table, smx = softmax table [V][d] read in op dtype."""
        if self.shared_embedding:          # one table read as source / target embedding (fp32) and softmax (op dtype)
            specs = [("embedding/shared/kernel", "shr", ("emb_src",))]
        else:
            specs = [("embedding/src_embedding/kernel", "tab", ("emb_src",)),
                     ("embedding/dst_embedding/kernel", "tab", ("emb_dst",)),
                     ("embedding/softmax/kernel", "smx", ("softmax_w",))]

        def dense(name, path):
            specs.append((name + "/kernel", "mat", path + ("w",)))
            specs.append((name + "/bias", "vec", path + ("b",)))

        def lnorm(name, path):
            specs.append((name + "/layer_norm_scale", "vec", path + ("g",)))
            specs.append((name + "/layer_norm_bias", "vec", path + ("b",)))

        for i in range(self.num_blocks):
            p, e = "encoder/block_%d/" % i, ("enc", i)
            dense(p + "encoder_self_attention/qkv_transform_single", e + ("qkv",))
            dense(p + "encoder_self_attention/output_transform_single", e + ("o",))
            dense(p + "conv_hidden_relu/conv1_single", e + ("ffn1",))
            dense(p + "conv_hidden_relu/conv2_single", e + ("ffn2",))
            lnorm(p + "layer_norm", e + ("ln0",))
            lnorm(p + "layer_norm_1", e + ("ln1",))
            p, b = "decoder/block_%d/" % i, ("dec", i)
            dense(p + "decoder_self_attention/qkv_transform_single", b + ("qkv",))
            dense(p + "decoder_self_attention/output_transform_single", b + ("o",))
            dense(p + "decoder_vanilla_attention/q_transform_single", b + ("cq",))
            dense(p + "decoder_vanilla_attention/kv_transform_single", b + ("ckv",))
            dense(p + "decoder_vanilla_attention/output_transform_single", b + ("co",))
            dense(p + "conv_hidden_relu/conv1_single", b + ("ffn1",))
            dense(p + "conv_hidden_relu/conv2_single", b + ("ffn2",))
            lnorm(p + "layer_norm", b + ("ln0",))
            lnorm(p + "layer_norm_1", b + ("ln1",))
            lnorm(p + "layer_norm_2", b + ("ln2",))
        return specs

    def _enable_training(self):
        """Re-homes the weights for the update step: ONE flat fp32 master buffer (every tensor in its packed
        layout at a 256-byte aligned offset), one flat gradient buffer and one flat RMSProp accumulator with the
        same offsets - so the optimizer is one launch and the multi-GPU gradient average one RCCL all-reduce -
        plus, for a bf16 model, a flat bf16 operand copy the optimizer refreshes.  The nmtgan_generator struct is
        re-pointed into these buffers; a second struct with the same shape names the gradients."""
        if self._train is not None:
            return self._train
        W = self._W_src
        if W is None:
            raise RuntimeError("Model: init_and_restore()/load_weights() has not been called")
        dev = self.device
        specs, chunks, shapes, off = self._param_specs(), [], {}, 0
        for name, kind, path in specs:
            a = np.asarray(W[name], np.float32)
            shapes[name] = a.shape
            if kind == "mat":
                a = a.reshape(a.shape[-2], a.shape[-1]).T            # Wt[N][K]
            a = np.ascontiguousarray(a).reshape(-1)
            chunks.append((name, kind, path, off, a))
            off += (a.size + 63) // 64 * 64
        host = np.zeros(off, np.float32)
        for _, _, _, o, a in chunks:
            host[o:o + a.size] = a
        master = torch.from_numpy(host).to(dev)
        grad = torch.zeros(off, dtype=torch.float32, device=dev)
        ms = torch.ones(off, dtype=torch.float32, device=dev)        # RMSProp `rms` slot starts at one
        # operand copies: `opc` in the model's own op dtype (what get_reward / generate_step read); an fp16 model
        # runs its update step on a bf16 copy (`opc_train`): the backward pass keeps d(logits)-sized values in the
        # operand type and those underflow in fp16
        opc = master.to(self._tdtype) if self.precision not in ("fp32", _lib.SPLIT_PRECISION) else None
        opc_train = master.to(torch.bfloat16) if self.precision in ("fp16", _lib.SPLIT_PRECISION) else opc
        # split tier: the scoring struct reads a flat fp16 buffer with every matrix in its [N][3K] split form at three
        # times its master offset (refreshed from the master after each update), the update step runs on the bf16 copy
        spl = torch.zeros(3 * off, dtype=torch.float16, device=dev) if self.split else None
        old = self._c

        def put(struct, path, addr):
            obj = struct
            for k in path[:-1]:
                obj = obj[k] if isinstance(k, int) else getattr(obj, k)
            setattr(obj, path[-1], addr)

        def build(opbuf, opcode):
            c = _lib.Generator()
            for f in ("op_dtype", "d", "heads", "n_blocks", "vocab_src", "vocab_dst", "emb_mult", "timing", "max_pos"):
                setattr(c, f, getattr(old, f))
            c.op_dtype = opcode
            for name, kind, path, o, a in chunks:
                use_op = opbuf is not None and kind in ("mat", "smx")
                put(c, path, (opbuf.data_ptr() + 2 * o) if use_op else (master.data_ptr() + 4 * o))
                if kind == "shr":          # the three roles of the shared table
                    c.emb_dst = c.emb_src
                    c.softmax_w = (opbuf.data_ptr() + 2 * o) if opbuf is not None else c.emb_src
            return c

        c = build(opc, self._op)
        c_train = build(opc_train, _lib.NMTGAN_BF16) if self.precision in ("fp16", _lib.SPLIT_PRECISION) else c
        if self.split:
            c.gemm_split = 1
            for name, kind, path, o, a in chunks:
                if kind in ("mat", "smx"):
                    put(c, path, spl.data_ptr() + 2 * 3 * o)
                elif kind == "shr":
                    c.softmax_w = spl.data_ptr() + 2 * 3 * o
        gc = _lib.Generator()
        for f in ("op_dtype", "d", "heads", "n_blocks", "vocab_src", "vocab_dst", "emb_mult", "timing", "max_pos"):
            setattr(gc, f, getattr(old, f))
        index = {}
        for name, kind, path, o, a in chunks:
            put(gc, path, grad.data_ptr() + 4 * o)
            if kind == "shr":              # ONE gradient buffer accumulates the three roles of the shared table
                gc.emb_dst = gc.softmax_w = gc.emb_src
            index[name] = (kind, o, a.size)
        timing = [t for t in self._keep if t.data_ptr() == old.timing]
        self._keep = timing + [master, grad, ms] + [t for t in (opc, opc_train, spl) if t is not None]
        self._c = c
        self._train = dict(master=master, grad=grad, ms=ms, op=opc, op_train=opc_train, c_train=c_train,
                           op_train_code=_lib.NMTGAN_BF16 if self.precision in ("fp16", _lib.SPLIT_PRECISION) else self._op,
                           gc=gc, index=index, shapes=shapes, n=off, split=spl,
                           loss=torch.zeros(1, dtype=torch.float32, device=dev), steps=0)
        self._refresh_split()
        return self._train

    def _refresh_split(self):
        """Split tier: rebuild the [N][3K] fp16 form of every matrix and of the softmax table from the fp32 master."""
        tr = self._train
        if tr is None or tr.get("split") is None:
            return
        master, spl = tr["master"], tr["split"]
        for name, (kind, o, n) in tr["index"].items():
            if kind in ("mat", "smx", "shr"):
                shp = tr["shapes"][name]
                rows = shp[-1] if kind == "mat" else shp[0]        # Wt[N][K]: N = dout; tables are [V][d]
                w = master[o:o + n].view(rows, n // rows)
                spl[3 * o:3 * o + 3 * n].view(rows, 3 * (n // rows)).copy_(_lib.split3_weights(w))

    def get_weights(self) -> Dict[str, np.ndarray]:
        """Current trainable variables under their TF names and shapes (what tf.train.Saver would write)."""
        if self._train is None:
            return {name: np.asarray(self._W_src[name], np.float32) for name, _, _ in self._param_specs()}
        tr = self._train
        host = tr["master"].cpu().numpy()
        out = {}
        for name, (kind, o, n) in tr["index"].items():
            a = host[o:o + n]
            shp = tr["shapes"][name]
            if kind == "mat":
                a = a.reshape(shp[-1], shp[-2]).T
            out[name] = np.ascontiguousarray(a.reshape(shp))
        return out

    def _restore_from(self, path):
        self.init_and_restore(modelFile=path)

    def save(self, prefix):
        """saver.save(sess, prefix) (gan_train.py:118): a TF V2 bundle of the trainable variables."""
        from . import tf_checkpoint
        tf_checkpoint.save_checkpoint(str(prefix), self.get_weights())

    def generate_step_and_update(self, sentence_x, sentence_y, reward, *, return_grads=False):
        """Reference: model.py:273-276 -> the loss half of build_generate (model.py:178-190): teacher-forced
        decoder over shift_right(Y), gan_output loss -sum p(y_t) reward_t / B (model.py:649-666), gradients of
        every trainable variable, RMSProp(config.generator.learning_rate) or Adadelta() update.  Returns the loss.
        With torch.distributed initialised the sentences are split across ranks like the reference's towers
        (split_tensor), every rank's loss is normalised by its own batch and gradients are averaged
        (average_gradients, model.py:689-722) with one RCCL all-reduce of the flat gradient buffer."""
        x = np.ascontiguousarray(np.asarray(sentence_x), dtype=np.int32)
        y = np.ascontiguousarray(np.asarray(sentence_y), dtype=np.int32)
        r = np.ascontiguousarray(np.asarray(reward), dtype=np.float32)
        if x.ndim != 2 or y.ndim != 2 or x.shape[0] != y.shape[0] or r.shape != y.shape:
            raise ValueError("generate_step_and_update: x [B, Ls], y [B, T] and reward [B, T] expected")
        world, rank = 1, 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            world, rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
        bounds = split_bounds(x.shape[0], world)
        b0, b1 = bounds[rank], bounds[rank + 1]
        if b1 <= b0:
            raise ValueError("generate_step_and_update: batch smaller than the number of ranks")
        x, y, r = x[b0:b1], y[b0:b1], r[b0:b1]
        B, Ls = x.shape
        T = y.shape[1]
        tr = self._enable_training()
        g = tr["c_train"]
        xd = self._dev_i32(x)
        pad = (xd == 0).to(torch.uint8).contiguous()
        yd, dec_in = self._dev_i32(y), self._dev_i32(shift_right(y))
        rd = torch.as_tensor(r).to(self.device)
        ws = self._workspace(lib.nmtgan_pg_workspace_bytes(C.byref(g), B, Ls, T))
        st = _lib.stream_ptr()
        check(lib.nmtgan_pg_grads(C.byref(g), C.byref(tr["gc"]), ptr(xd), ptr(pad), ptr(yd), ptr(dec_in), ptr(rd), B, Ls, T,
                                  ptr(tr["loss"]), None, ptr(ws), ws.numel(), st), "nmtgan_pg_grads")
        loss = tr["loss"].clone()
        if world > 1:
            work_g, work_l = tr["grad"], loss
            if torch.distributed.get_backend() != "nccl":
                work_g, work_l = tr["grad"].cpu(), loss.cpu()
            torch.distributed.all_reduce(work_g)
            torch.distributed.all_reduce(work_l)
            if work_g is not tr["grad"]:
                tr["grad"].copy_(work_g)
            loss = work_l.to(self.device) / world                     # tf.reduce_mean(loss_list), model.py:187
        grads = None
        if return_grads:
            host = (tr["grad"] / world).cpu().numpy()
            grads = {}
            for name, (kind, o, n) in tr["index"].items():
                a, shp = host[o:o + n], tr["shapes"][name]
                grads[name] = np.ascontiguousarray((a.reshape(shp[-1], shp[-2]).T if kind == "mat" else a).reshape(shp))
        if self.g_optimizer == 'adadelta':             # tf.train.AdadeltaOptimizer(): its own defaults, model.py:115-116
            if "accum" not in tr:
                tr["accum"], tr["accum_update"] = torch.zeros_like(tr["master"]), torch.zeros_like(tr["master"])
            check(lib.nmtgan_adadelta_update(ptr(tr["master"]), ptr(tr["accum"]), ptr(tr["accum_update"]), ptr(tr["grad"]),
                                             tr["n"], 0.001, 0.95, 1e-8, 1.0 / world, 0.0, ptr(tr["op_train"]),
                                             tr["op_train_code"], st), "nmtgan_adadelta_update")
        else:
            lr = float(_cfg_get(self.config, "generator.learning_rate", 1e-5))
            check(lib.nmtgan_rmsprop_update(ptr(tr["master"]), ptr(tr["ms"]), None, ptr(tr["grad"]), tr["n"], lr, 0.9, 0.0,
                                            1e-10, 1.0 / world, 0.0, ptr(tr["op_train"]), tr["op_train_code"], st),
                  "nmtgan_rmsprop_update")
        if tr["op"] is not None and tr["op"] is not tr["op_train"]:        # fp16 scoring copy from the updated master
            check(lib.nmtgan_clip_values(ptr(tr["master"]), tr["n"], float("-inf"), float("inf"), ptr(tr["op"]), self._op, st),
                  "nmtgan_clip_values")
        self._refresh_split()
        tr["steps"] += 1
        torch.cuda.synchronize(self.device)
        out = float(loss.item())
        return (out, grads) if return_grads else out

    # ------------------------------------------------------------------ parity hook
    def teacher_forced_logits(self, x, y, *, device_out=False):
        """Logits [B, T, V] of decoder(shift_right(y)) given x (model.py:486-618, 804-820); device_out=True
        returns the device tensor instead of a numpy copy (3 GB at 512 sentences x 50 x 32000)."""
        x = np.ascontiguousarray(np.asarray(x), dtype=np.int32)
        y = np.ascontiguousarray(np.asarray(y), dtype=np.int32)
        B, Ls = x.shape
        T = y.shape[1]
        xd = self._dev_i32(x)
        pad = (xd == 0).to(torch.uint8).contiguous()
        dec_in = self._dev_i32(shift_right(y))
        out = torch.empty((B, T, self.dst_vocab_size), dtype=torch.float32, device=self.device)
        g = self.c_struct
        ws = self._workspace(lib.nmtgan_teacher_forced_workspace_bytes(C.byref(g), B, Ls, T))
        check(lib.nmtgan_teacher_forced_logits(C.byref(g), ptr(xd), ptr(pad), ptr(dec_in), B, Ls, T, ptr(out),
                                               ptr(ws), ws.numel(), _lib.stream_ptr()), "nmtgan_teacher_forced_logits")
        torch.cuda.synchronize(self.device)
        return out if device_out else out.cpu().numpy()

    # ------------------------------------------------------------------ the hot path
    def make_uniforms(self, rollnum, max_len, batch, seed):
        """u[R, T-1, B, T] ~ U[0,1) fp32 from torch's Philox generator on the device."""
        gen = torch.Generator(device=self.device)
        gen.manual_seed(int(seed))
        return torch.rand((rollnum, max_len - 1, batch, max_len), generator=gen, device=self.device,
                          dtype=torch.float32)

    def get_reward(self, x, x_to_maxlen, y_sample, y_sample_mask, rollnum, disc, max_len=50, bias_num=None,
                   data_util=None, *, uniforms=None, seed=None, return_details=False, shard=True):
        """Reference: Model.get_reward, model.py:291-346.  Returns float32 [B, max_len].

        uniforms: optional [rollnum, max_len-1, B, max_len] array/tensor, u[r, g-1, b, t] is consumed
        when rollout (r, give_num g) of sentence b samples position t >= g; else drawn from `seed`.
        """
        x = np.ascontiguousarray(np.asarray(x), dtype=np.int32)
        x_pad_T = np.ascontiguousarray(np.asarray(x_to_maxlen), dtype=np.int32)
        y = np.ascontiguousarray(np.asarray(y_sample), dtype=np.int32)
        mask = np.ascontiguousarray(np.asarray(y_sample_mask), dtype=np.float32)
        B, Ls = x.shape
        T = int(max_len)
        if y.shape != (B, T) or mask.shape != (B, T) or x_pad_T.shape != (B, T):
            raise ValueError("x_to_maxlen, y_sample and y_sample_mask must be [batch, max_len=%d]" % T)
        if Ls > T:
            raise ValueError("source longer than max_len (extend_sentence_to_maxlen would raise, quirk B.5)")
        if disc.max_len != T:
            raise ValueError("disc.max_len != max_len")
        rollnum = int(rollnum)
        if uniforms is None:
            if seed is None:           # fresh noise per call, as tf.multinomial draws it (model.py:239)
                seed = int(self.reward_seed_base) + self._reward_calls
            uniforms = self.make_uniforms(rollnum, T, B, seed)
        self._reward_calls += 1
        u = torch.as_tensor(uniforms)
        if tuple(u.shape) != (rollnum, T - 1, B, T):
            raise ValueError("uniforms must have shape [rollnum, max_len-1, batch, max_len]")
        u = u.to(device=self.device, dtype=torch.float32).contiguous()

        world, rank = 1, 0
        if shard and torch.distributed.is_available() and torch.distributed.is_initialized():
            world, rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
        ct = self.conv_form(mask, rollnum, T, disc)
        if world > 1 and B < world:
            out = self._reward_roll_sharded(x, x_pad_T, y, mask, rollnum, disc, T, bias_num, u, world, rank, ct)
            torch.cuda.synchronize(self.device)
            return out.cpu().numpy()
        bounds = split_bounds(B, world)
        b0, b1 = bounds[rank], bounds[rank + 1]
        local = self._reward_local(x[b0:b1], x_pad_T[b0:b1], y[b0:b1], mask[b0:b1], rollnum, disc, T, bias_num, u,
                                   b_offset=b0, B_global=B, return_details=return_details, conv_table=ct)
        if world == 1:
            out = local
        else:
            out = gather_rows(local, bounds, self.device, self._gather_bufs, copy=False)    # copied to the host below
        torch.cuda.synchronize(self.device)
        return out.cpu().numpy()

    def conv_form(self, mask, rollnum, T, disc):
        """Which form of the rollouts' discriminator conv a call takes, as nmtgan_reward_plan.conv_table (> 0 table form,
        < 0 conv GEMMs), decided ONCE from the whole batch: the two forms round differently, so every shard of a job must
        take the form the unsharded call takes for its scores to be the same bits.  The rule is the library's default -
        the table form when the rollouts hold at least four tokens per vocabulary entry, never for the fp32 tier -
        applied to the global row count; Model.conv_table != 0 overrides it."""
        if int(self.conv_table) != 0:
            return int(self.conv_table)
        lens = (np.asarray(mask) != 0).sum(axis=1)
        n_rows = int(rollnum) * int(np.clip(lens - 1, 0, None).sum())
        return 1 if (getattr(disc, "precision", None) != "fp32" and n_rows * int(T) >= 4 * int(disc.vocab_size)) else -1

    def _reward_roll_sharded(self, x, x_pad_T, y, mask, rollnum, disc, T, bias_num, u, world, rank, conv_table=None):
        """Fewer sentences than ranks: the (sentence, roll) units are cut across the ranks (all prefixes of a
        unit stay on one rank).  A rank scores its pieces - one sentence with a range of rolls each, uniforms
        addressed by global sentence and roll - and leaves P(real) of every rollout in a [units, T] block
        (column g-1 = prefix g, last column = P(real) of y_sample itself); one all-gather of the blocks, then
        every rank runs the reward reduction over all rolls in roll order: the sums are those of the unsharded
        call.  Replaces the tower split/concat of model.py:248-261 for B < world."""
        B, R, dev = x.shape[0], rollnum, self.device
        u0, u1, pieces = roll_shard_pieces(B, R, world, rank)
        part = torch.zeros((max(u1 - u0, 0), T), dtype=torch.float32, device=dev)
        for b, r0, r1 in pieces:
            prep = self.prepare_reward(x[b:b + 1], x_pad_T[b:b + 1], y[b:b + 1], mask[b:b + 1], r1 - r0, disc, T, None, u,
                                       b_offset=b, B_global=B, r_offset=r0, return_details="scores", conv_table=conv_table)
            prep.launch()
            plan, d = prep.plan, prep.details
            rows = torch.arange(r0, r1, device=dev) + (b * R - u0)
            part[rows, T - 1] = d["p_base"][0]
            if plan.n_rows > 0:
                rr = torch.as_tensor(plan.row_r.astype(np.int64), device=dev) + (b * R + r0 - u0)
                gg = torch.as_tensor(plan.row_g.astype(np.int64), device=dev) - 1
                part[rr, gg] = d["p_rows"][:plan.n_rows]
            torch.cuda.synchronize(dev)        # the next piece re-uses the workspace and the score buffers' owner
        full = gather_rows(part, split_bounds(B * R, world), dev, self._gather_bufs)          # [B*R, T]
        p = full[:, :T - 1].reshape(B, R, T - 1).permute(0, 2, 1).contiguous()                 # [B, T-1, R]
        p_base = full[:, T - 1].reshape(B, R)[:, 0].contiguous()
        lens = (mask != 0).sum(axis=1)
        first = (np.arange(B)[:, None] * (T - 1) + np.arange(T - 1)[None, :]) * R
        grp_first = np.where(np.arange(1, T)[None, :] < lens[:, None], first, -1).astype(np.int32)
        gf = self._dev_i32(grp_first)
        md = torch.as_tensor(np.ascontiguousarray(mask, dtype=np.float32)).to(dev)
        rewards = torch.zeros((B, T), dtype=torch.float32, device=dev)
        check(lib.nmtgan_reward_reduce(ptr(p), ptr(p_base), ptr(gf), ptr(md), B, T, R, 0 if bias_num is None else 1,
                                       0.0 if bias_num is None else float(bias_num), ptr(rewards), _lib.stream_ptr()),
              "nmtgan_reward_reduce")
        return rewards

    def _reward_local(self, x, x_pad_T, y, mask, rollnum, disc, T, bias_num, u, b_offset, B_global, return_details,
                      conv_table=None):
        if x.shape[0] == 0:
            return torch.zeros((0, T), dtype=torch.float32, device=self.device)
        prep = self.prepare_reward(x, x_pad_T, y, mask, rollnum, disc, T, bias_num, u, b_offset=b_offset,
                                   B_global=B_global, return_details=return_details, conv_table=conv_table)
        rewards = prep.launch()
        if return_details:
            torch.cuda.synchronize(self.device)
            self.last_details = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in prep.details.items()}
        return rewards

    def prepare_reward(self, x, x_pad_T, y, mask, rollnum, disc, T, bias_num, u, b_offset=0, B_global=None,
                       return_details=False, own_workspace=False, stream=None, r_offset=0, conv_table=None):
        """Host half of get_reward: build the rollout plan, upload ids/tables, size the workspace.
        Returns a PreparedReward whose launch() only enqueues device work (inputs resident in HBM).
        conv_table: the plan field of that name (None: Model.conv_table, 0 = the library's rule on THIS shard's rows;
        get_reward passes the choice it made for the whole batch, see conv_form)."""
        dev = self.device
        check_pairing(self.precision, getattr(disc, "precision", None))
        x = np.ascontiguousarray(np.asarray(x), dtype=np.int32)
        B, Ls = x.shape
        plan = build_rollout_plan(mask, rollnum, b_offset=b_offset, B_global=B if B_global is None else B_global,
                                  r_offset=r_offset)
        xd = self._dev_i32(x)
        t = {
            "x": xd, "x_pad": (xd == 0).to(torch.uint8).contiguous(), "x_pad_T": self._dev_i32(x_pad_T),
            "y": self._dev_i32(y), "dec_in": self._dev_i32(shift_right(y)),
            "mask": torch.as_tensor(np.ascontiguousarray(mask, dtype=np.float32)).to(dev),
            "row_b": self._dev_i32(plan.row_b), "row_g": self._dev_i32(plan.row_g),
            "row_gp1": self._dev_i32(plan.row_g + 1), "row_lrow": self._dev_i32(plan.row_lrow),
            "row_uoff": torch.as_tensor(plan.row_uoff).to(dev), "grp_first": self._dev_i32(plan.grp_first),
            "step_off": torch.as_tensor(plan.step_off).to(dev),
            "u": torch.as_tensor(u).to(device=dev, dtype=torch.float32).contiguous(),
            "rewards": torch.zeros((B, T), dtype=torch.float32, device=dev),
        }
        grp_live = np.ascontiguousarray(plan.grp_live, dtype=np.int32)
        if self.group_slots and plan.grp_slots > 0 and plan.grp_table.shape[0] > 0:
            t["grp_table"] = self._dev_i32(plan.grp_table)
        live = np.ascontiguousarray(plan.live, dtype=np.int32)
        step_off_h = np.ascontiguousarray(plan.step_off, dtype=np.int64)
        row_g_h = np.ascontiguousarray(plan.row_g, dtype=np.int32)
        p = _lib.RewardPlan()
        p.B, p.Ls, p.T, p.rollnum = B, Ls, T, int(rollnum)
        p.n_rows, p.n_steps = plan.n_rows, plan.n_steps
        p.live = live.ctypes.data
        p.step_off_host = step_off_h.ctypes.data
        p.step_off = ptr(t["step_off"])
        p.row_g_host = row_g_h.ctypes.data if plan.n_rows > 0 else None
        p.x, p.x_pad, p.x_pad_to_T = ptr(t["x"]), ptr(t["x_pad"]), ptr(t["x_pad_T"])
        p.y, p.dec_in, p.mask = ptr(t["y"]), ptr(t["dec_in"]), ptr(t["mask"])
        p.row_b, p.row_g, p.row_gp1 = ptr(t["row_b"]), ptr(t["row_g"]), ptr(t["row_gp1"])
        p.row_lrow, p.row_uoff, p.grp_first = ptr(t["row_lrow"]), ptr(t["row_uoff"]), ptr(t["grp_first"])
        p.uniforms = ptr(t["u"])
        p.use_bias = 0 if bias_num is None else 1
        p.bias_num = 0.0 if bias_num is None else float(bias_num)
        p.logits_chunk_rows = self.logits_chunk_rows
        p.disc_chunk_rows = self.disc_chunk_rows
        p.rewards = ptr(t["rewards"])
        p.conv_bucket_taps = int(self.conv_bucket_taps)
        p.conv_table = int(self.conv_table if conv_table is None else conv_table)
        p.conv_table_max_bytes = int(self.conv_table_max_bytes)
        if "grp_table" in t:
            p.grp_table, p.grp_slots, p.grp_live = ptr(t["grp_table"]), int(plan.grp_slots), grp_live.ctypes.data
        if self.overlap:
            if self._aux_stream is None:
                self._aux_stream = torch.cuda.Stream(device=dev)
            p.aux_stream = self._aux_stream.cuda_stream
            p.aux_gemm_wgs, p.aux_min_rows = int(self.aux_gemm_wgs), int(self.aux_min_rows)
        details = None
        if return_details:             # "scores": only P(real) per rollout and of y_sample
            details = {
                "plan": plan,
                "p_rows": torch.zeros(max(plan.n_rows, 1), dtype=torch.float32, device=dev),
                "p_base": torch.zeros(B, dtype=torch.float32, device=dev),
            }
            p.p_rows, p.p_base = ptr(details["p_rows"]), ptr(details["p_base"])
            if return_details != "scores":
                details["rollouts"] = torch.zeros((max(plan.n_rows, 1), T), dtype=torch.int32, device=dev)
                details["prefix_logits"] = torch.zeros((B, T, self.dst_vocab_size), dtype=torch.float32, device=dev)
                p.rollouts, p.prefix_logits = ptr(details["rollouts"]), ptr(details["prefix_logits"])
        g, dsc = self.c_struct, disc.c_struct
        need = lib.nmtgan_rollout_reward_workspace_bytes(C.byref(g), C.byref(dsc), C.byref(p))
        if own_workspace:              # several prepared calls alive at once (sub-batches on their own streams)
            if need == 0:
                raise _lib.NmtGanError("workspace query failed: " + lib.nmtgan_last_error().decode())
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            if self.overlap:
                keep_aux = torch.cuda.Stream(device=dev)
                p.aux_stream = keep_aux.cuda_stream
                t["_aux_stream"] = keep_aux
        else:
            ws = self._workspace(need)
        return PreparedReward(self, disc, plan, p, t, (live, step_off_h, row_g_h, grp_live), ws, details, stream)

    def get_reward_Obinforced(self, *a, **k):
        raise NotImplementedError("get_reward_Obinforced is never called by the reference drivers")


def check_pairing(gen_precision, disc_precision):
    """The tiers that may score together (mirrors the library's plan check): a split-precision generator (fp32
    storage, fp16 products) pairs with a 16-bit discriminator - plain fp16 / bf16 or the split form 'fp16x3' -, every
    other generator with a discriminator of its own operand type ('fp16' also with the split-form discriminator, whose
    operands are fp16).  In particular Model('fp16x3') + DisCNN('fp32') is refused: the two would not share an
    operand type anywhere and the library has no such path."""
    if disc_precision is None:
        return
    ok = {_lib.SPLIT_PRECISION: ("fp16", "bf16", _lib.SPLIT_PRECISION), "fp16": ("fp16", _lib.SPLIT_PRECISION),
          "bf16": ("bf16",), "fp32": ("fp32",)}
    if disc_precision not in ok.get(gen_precision, ()):
        raise ValueError("Model(precision=%r) cannot score with DisCNN(precision=%r); allowed: %s"
                         % (gen_precision, disc_precision, ", ".join(ok.get(gen_precision, ()))))


class PreparedReward(object):
    """One get_reward call with every input already resident in HBM; launch() enqueues the whole
    path on the current stream (no allocation, no synchronisation) and returns the device tensor
    [B, T] the rewards are written to."""

    def __init__(self, model, disc, plan, cplan, tensors, host_tables, ws, details, stream=None):
        self.model, self.disc, self.plan, self.cplan = model, disc, plan, cplan
        self.tensors, self.host_tables, self.ws, self.details = tensors, host_tables, ws, details
        self.stream = stream           # torch.cuda.Stream to enqueue on (None: the current stream)

    @property
    def workspace_bytes(self):
        return int(self.ws.numel())

    def with_schedule(self, overlap):
        """Same prepared call (same tensors, workspace and plan) on the other schedule: overlap=False puts
        the whole path on the current stream, True hands the discriminator to the model's auxiliary stream."""
        cplan = type(self.cplan).from_buffer_copy(self.cplan)   # pointers still borrowed from self
        if overlap:
            if self.model._aux_stream is None:
                self.model._aux_stream = torch.cuda.Stream(device=self.model.device)
            cplan.aux_stream = self.model._aux_stream.cuda_stream
        else:
            cplan.aux_stream = None
        return PreparedReward(self.model, self.disc, self.plan, cplan, self.tensors, self.host_tables, self.ws,
                              self.details, self.stream)

    def capture(self):
        """Records the whole call - every kernel, memset and the fork/join of the two streams - into ONE HIP graph
        (the library only enqueues: no allocation, no synchronisation, no host callback); replay() then launches
        it with a single call.  One un-captured launch runs first so that lazily created state (event pool,
        function attributes) exists before the capture."""
        self.launch()
        torch.cuda.synchronize(self.model.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="relaxed"):
            self.launch()
        self._graph = graph
        return self

    def replay(self):
        self._graph.replay()
        return self.tensors["rewards"]

    def launch(self):
        g, dsc = self.model.c_struct, self.disc.c_struct
        st = _lib.stream_ptr() if self.stream is None else self.stream.cuda_stream
        check(lib.nmtgan_rollout_reward(C.byref(g), C.byref(dsc), C.byref(self.cplan), ptr(self.ws), self.ws.numel(),
                                        st), "nmtgan_rollout_reward")
        return self.tensors["rewards"]


def gather_rows(local: torch.Tensor, bounds, device, cache=None, copy=True) -> torch.Tensor:
    """All-gather of per-rank reward rows [B_i, T] -> [B, T] (the implicit tf.concat of the tower
    outputs, model.py:261 / cnn_discriminator.py:558).  Shards may be uneven (split_tensor uses
    float linspace boundaries), so rows are padded to the largest shard for the collective.
    `cache` (a dict the caller keeps) holds the send/receive buffers between calls, so a steady-state
    call allocates only its result.  copy=False returns, where it can (even shards under RCCL), a VIEW of the cached
    receive buffer: no allocation at all, valid until the next call with the same cache - for callers that consume
    the rows at once (get_reward copies them to the host, bench.py only checks them).
    gloo gets host tensors (synchronous copies), RCCL device tensors."""
    import torch.distributed as dist
    world = dist.get_world_size()
    sizes = [bounds[i + 1] - bounds[i] for i in range(world)]
    mx = max(max(sizes), 1)
    T = local.shape[1]
    work_dev = device if dist.get_backend() == "nccl" else torch.device("cpu")
    key = (world, mx, T, str(work_dev))
    bufs = cache.get(key) if cache is not None else None
    if bufs is None:
        bufs = (torch.zeros((mx, T), dtype=torch.float32, device=work_dev),
                torch.zeros((world, mx, T), dtype=torch.float32, device=work_dev))
        if cache is not None:
            cache[key] = bufs
    padded, recv = bufs
    padded[: local.shape[0]].copy_(local)
    dist.all_gather([recv[i] for i in range(world)], padded)
    if all(n == mx for n in sizes):
        out = recv.reshape(world * mx, T).to(device)          # no copy when recv already lives on `device`
        return out.clone() if copy and out.data_ptr() == recv.data_ptr() else out
    return torch.cat([recv[i, : sizes[i]] for i in range(world)], dim=0).to(device)
