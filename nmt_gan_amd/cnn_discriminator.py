"""DisCNN facade: the inference surface of the reference's cnn_discriminator.DisCNN that
Model.get_reward drives (model.py:311-313), backed by libnmtgan_hip.so.

Inference: the two-tower multi-width CNN + highway + 2-way softmax of
build_discriminator_body (cnn_discriminator.py:440-522) with BatchNorm in moving-statistics mode.
Training: train_step() is one iteration of DisCNN.train (cnn_discriminator.py:664-691) on a
batch the caller supplies; the file iterators around it (train(), :593-704) are data pipeline and stay out.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr

BN_EPS = 1e-3          # tf.contrib.layers.batch_norm default epsilon (cnn_discriminator.py:22-33)


_op_torch_dtype = _lib.op_dtype_of


class DisCNN(object):
    """Same constructor signature as the reference (cnn_discriminator.py:117-119); the data-path,
    optimizer and training arguments are accepted and ignored.  `precision`/`device` are
    build-only keyword extensions; the default precision is the split-precision tier 'fp16x3', the fastest one whose
    P(real) stays within 1e-3 of the reference's fp32 arithmetic on every weight set tested ('fp16' is the throughput
    tier: 3x cheaper conv, P(real) off by up to 2.5e-2 where the 2-way score is large)."""

    # names of the reference placeholders / fetch (cnn_discriminator.py:532-558): kept as feed keys
    dis_input_x = "dis_input_x"
    dis_input_xs = "dis_input_xs"
    dis_dropout_keep_prob = "dis_dropout_keep_prob"
    dis_ypred_for_auc = "dis_ypred_for_auc"

    def __init__(self, sess, max_len, num_classes, vocab_size, batch_size, dim_word, filter_sizes, num_filters,
                 source_dict=None, target_dict=None, gpu_device=None, positive_data=None, negative_data=None,
                 source_data=None, vocab_size_s=None, dev_positive_data=None, dev_negative_data=None,
                 dev_source_data=None, max_epoches=10, dispFreq=1, saveFreq=10, devFreq=1000, clip_c=1.0,
                 optimizer='adadelta', saveto='discriminator', reload=False, reshuffle=False, l2_reg_lambda=0.0,
                 scope='discnn', init_device="/cpu:0", reuse_var=False, precision="fp16x3", device="cuda:0",
                 chunk_rows=4096):
        if num_classes != 2:
            raise ValueError("the reward path reads P(real) of a 2-way softmax (model.py:315); num_classes must be 2")
        self.sess = sess
        self.max_len = int(max_len)
        self.num_classes = num_classes
        self.vocab_size = int(vocab_size)
        self.vocab_size_s = int(vocab_size if vocab_size_s is None else vocab_size_s)
        self.batch_size = batch_size
        self.dim_word = int(dim_word)
        self.filter_sizes = [int(w) for w in filter_sizes]
        self.num_filters = [int(n) for n in num_filters]
        self.num_filters_total = int(sum(self.num_filters))
        self.scope = scope
        self.saveto = saveto
        # 'fp16x3': the split-precision scoring form (nmtgan_discriminator.gemm_split) - every product of the two towers
        # as one fp16 MFMA GEMM over hi/lo operand splits, fp32-class P(real) at three times the fp16 conv cost.  The
        # reference scores in fp32 (cnn_discriminator.py:440-522) and its 2-way score reaches |s| ~ 100, where plain
        # 16-bit operands move P(real) by percents; this is the tier that keeps rewards within 1e-3 of the fp32 path.
        self.precision = precision
        self.split = precision == _lib.SPLIT_PRECISION
        self.split_ablate = {}     # error-budget runs only: stages packed WITHOUT their lo halves (see load_weights)
        self.device = torch.device(device)
        self.chunk_rows = int(chunk_rows)
        self._tdtype, self._op = _op_torch_dtype("fp16" if self.split else precision)
        if len(self.filter_sizes) != len(self.num_filters) or len(self.filter_sizes) > _lib.MAX_WIDTHS:
            raise ValueError("filter_sizes/num_filters mismatch or more than %d widths" % _lib.MAX_WIDTHS)
        if max(self.filter_sizes) > self.max_len:
            raise ValueError("filter width exceeds max_len")
        self.Fp = (self.num_filters_total + 63) // 64 * 64
        self._keep = []            # device tensors kept alive for the C struct
        self._c = None
        self._ws = None
        self._W_src, self._train = None, None
        self.optimizer_name = optimizer          # cnn_discriminator.py:153-166; RMSPropOptimizer(0.0001) for 'rmsprop'
        self.learning_rate = 1e-4
        from .model import _Saver
        self.saver = _Saver(self, "save", "restore")      # gan_train.py:161: discriminator.saver.save(sess, saveto)
        self.reload = bool(reload)
        if self.reload:                # cnn_discriminator.py:209-217: saver.restore(sess, saveto) in the constructor
            self.restore(saveto)

    # ------------------------------------------------------------------ weights
    def restore(self, prefix=None):
        """tf.train.Saver(params in scope).restore (cnn_discriminator.py:202-217): every variable of the
        scope, BatchNorm moving statistics included, from the TF V2 checkpoint prefix (default `saveto`)."""
        from . import tf_checkpoint
        return self.load_weights(tf_checkpoint.LazyCheckpoint(str(self.saveto if prefix is None else prefix)))

    def load_weights(self, W: Dict[str, np.ndarray]):
        """Pack TF-named variables (discnn/model_self/..., discnn/conv_maxpool-w/...)
        for the kernels: conv filters as Wt[n_w][w*E], conv bias + inference BN folded into
        scale/shift, highway matrices transposed and zero padded to Fp."""
        dev, td = self.device, self._tdtype
        s = self.scope + "/"
        E, F, Fp = self.dim_word, self.num_filters_total, self.Fp
        keep = []

        def dv(a, dtype):
            t = torch.as_tensor(np.ascontiguousarray(a)).to(device=dev, dtype=dtype).contiguous()
            keep.append(t)
            return t

        # split form: tables as [hi | lo | hi] rows, filters per tap and highway matrices as [hi | hi | lo].  An
        # ablated stage (error budget: split_ablate = {"emb": True, "conv": [widths], "pooled": True, "hw": True})
        # loses its lo halves, which leaves exactly the fp16 tier's rounding of that one operand: "emb" the sentence
        # image, "conv" the filters of the listed widths, "hw" the highway matrices, "pooled" the pooled features
        # (their lo block meets W_hi in the middle third of the highway image, so that third is zeroed)
        abl = self.split_ablate if self.split else {}

        def zero_third(t, third, period):          # zero one third of every `period`-wide group of columns
            v = t.view(t.shape[0], -1, 3, period // 3)
            v[:, :, third, :] = 0
            return t

        def pack_emb(emb):
            if not self.split:
                return dv(emb, td)
            t = _lib.split3_acts(torch.as_tensor(emb).to(dev))
            if abl.get("emb"):
                zero_third(t, 1, 3 * E)
            keep.append(t)
            return t

        def pack_conv(WcT, w):                     # Wt[n][w*E]
            if not self.split:
                return dv(WcT, td)
            t = _lib.split3_conv_weights(torch.as_tensor(np.ascontiguousarray(WcT)).to(dev), w)
            if w in (abl.get("conv") or ()):
                zero_third(t, 2, 3 * E)
            keep.append(t)
            return t

        def pack_hw(Wh):                           # Wt[Fp][Fp]
            if not self.split:
                return dv(Wh, td)
            t = _lib.split3_weights(torch.as_tensor(Wh).to(dev))
            if abl.get("hw"):
                zero_third(t, 2, 3 * Fp)
            if abl.get("pooled"):
                zero_third(t, 1, 3 * Fp)
            keep.append(t)
            return t

        c = _lib.Discriminator()
        c.op_dtype = self._op
        c.gemm_split = 1 if self.split else 0
        c.max_len, c.dim_word, c.vocab, c.vocab_s = self.max_len, E, self.vocab_size, self.vocab_size_s
        c.n_widths = len(self.filter_sizes)
        for i, (w, n) in enumerate(zip(self.filter_sizes, self.num_filters)):
            c.widths[i], c.n_filters[i] = w, n
        c.F, c.Fp = F, Fp

        def tower(ct, emb_name, conv_fmt, out_name, tr_name):
            emb = np.asarray(W[s + emb_name], np.float32)
            ct.emb = ptr(pack_emb(emb))
            for i, (w, n) in enumerate(zip(self.filter_sizes, self.num_filters)):
                p = s + conv_fmt % w + "/"
                Wc = np.asarray(W[p + "self_model/W"], np.float32).reshape(w * E, n)
                b = np.asarray(W[p + "self_model/b"], np.float32)
                bn = p + "conv_batch_norm/bn/"
                gamma, beta = np.asarray(W[bn + "gamma"], np.float32), np.asarray(W[bn + "beta"], np.float32)
                mean, var = np.asarray(W[bn + "moving_mean"], np.float32), np.asarray(W[bn + "moving_variance"], np.float32)
                inv = (np.float32(1.0) / np.sqrt(var + np.float32(BN_EPS))).astype(np.float32) * gamma
                shift = (b * inv + (beta - mean * inv)).astype(np.float32)
                ct.conv_w[i] = ptr(pack_conv(Wc.T, w))
                ct.conv_scale[i] = ptr(dv(inv, torch.float32))
                ct.conv_shift[i] = ptr(dv(shift, torch.float32))
            for field, name in (("hw_out", out_name), ("hw_tr", tr_name)):
                Wh = np.zeros((Fp, Fp), np.float32)
                Wh[:F, :F] = np.asarray(W[s + name + "/weights"], np.float32).T       # Wt[out][in]
                setattr(ct, field, ptr(pack_hw(Wh)))

        tower(c.tgt, "model_self/embeddingtable", "conv_maxpool-%d", "output_lin_0", "transform_lin_0")
        tower(c.src, "model_self/embeddingtable_s", "conv_s_maxpool-%d", "output_s_lin_0", "transform_s_lin_0")
        Wh = np.asarray(W[s + "model_self/W"], np.float32)                            # [2F, 2]
        head = np.zeros((2, 2 * Fp), np.float32)
        for cls in range(2):
            head[cls, :F] = Wh[:F, cls]
            head[cls, Fp:Fp + F] = Wh[F:, cls]
        c.head_w = ptr(dv(head, torch.float32))
        c.head_b = ptr(dv(np.asarray(W[s + "model_self/b"], np.float32), torch.float32))
        self._keep, self._c = keep, c
        self._W_src, self._train = W, None
        return self

    # ------------------------------------------------------------------ training (row f-4)
    def _tf_names(self):
        """(tower key, TF name prefix pieces) of both towers."""
        s = self.scope + "/"
        return (("tgt", s + "model_self/embeddingtable", s + "conv_maxpool-%d/", s + "output_lin_0/weights",
                 s + "transform_lin_0/weights"),
                ("src", s + "model_self/embeddingtable_s", s + "conv_s_maxpool-%d/", s + "output_s_lin_0/weights",
                 s + "transform_s_lin_0/weights"))

    def _enable_training(self):
        """Re-homes the weights for train_step(): one flat fp32 master buffer with every TRAINABLE tensor in its
        packed layout (tables [V][E], conv filters Wt[n][w*E], per-filter vectors concatenated over the widths,
        highway Wt[Fp][Fp] zero padded, head [2][2Fp]) at 256-byte aligned offsets, a flat gradient buffer and a flat
        RMSProp accumulator with the same offsets (one optimizer launch, one all-reduce), a flat bf16 operand copy for
        a bf16 model, and the BatchNorm moving statistics + the folded inference constants beside them."""
        if self._train is not None:
            return self._train
        W = self._W_src
        if W is None:
            raise RuntimeError("DisCNN: load_weights() has not been called")
        dev, E, F, Fp = self.device, self.dim_word, self.num_filters_total, self.Fp
        chunks, off = [], 0

        def add(key, arr, tf_name, kind):
            nonlocal off
            a = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
            chunks.append((key, tf_name, kind, off, a))
            off += (a.size + 63) // 64 * 64

        moving = {}
        for tkey, emb_name, conv_fmt, out_name, tr_name in self._tf_names():
            add((tkey, "emb"), np.asarray(W[emb_name], np.float32), emb_name, "tab")
            vecs = {k: [] for k in ("b", "gamma", "beta", "moving_mean", "moving_variance")}
            for w, n in zip(self.filter_sizes, self.num_filters):
                p = conv_fmt % w
                add((tkey, "conv_w", w), np.asarray(W[p + "self_model/W"], np.float32).reshape(w * E, n).T,
                    p + "self_model/W", "conv")
                vecs["b"].append(np.asarray(W[p + "self_model/b"], np.float32))
                for k in ("gamma", "beta", "moving_mean", "moving_variance"):
                    vecs[k].append(np.asarray(W[p + "conv_batch_norm/bn/" + k], np.float32))
            add((tkey, "conv_b"), np.concatenate(vecs["b"]), conv_fmt + "self_model/b", "vec")
            add((tkey, "gamma"), np.concatenate(vecs["gamma"]), conv_fmt + "conv_batch_norm/bn/gamma", "vec")
            add((tkey, "beta"), np.concatenate(vecs["beta"]), conv_fmt + "conv_batch_norm/bn/beta", "vec")
            moving[tkey] = (np.concatenate(vecs["moving_mean"]), np.concatenate(vecs["moving_variance"]))
            for key, name in (("hw_out", out_name), ("hw_tr", tr_name)):
                Wh = np.zeros((Fp, Fp), np.float32)
                Wh[:F, :F] = np.asarray(W[name], np.float32).T
                add((tkey, key), Wh, name, "hw")
        Wh = np.asarray(W[self.scope + "/model_self/W"], np.float32)
        head = np.zeros((2, 2 * Fp), np.float32)
        for cls in range(2):
            head[cls, :F] = Wh[:F, cls]
            head[cls, Fp:Fp + F] = Wh[F:, cls]
        add(("head_w",), head, self.scope + "/model_self/W", "head")
        add(("head_b",), np.asarray(W[self.scope + "/model_self/b"], np.float32), self.scope + "/model_self/b", "vec1")
        host = np.zeros(off, np.float32)
        for _, _, _, o, a in chunks:
            host[o:o + a.size] = a
        master = torch.from_numpy(host).to(dev)
        grad = torch.zeros(off, dtype=torch.float32, device=dev)
        ms = torch.ones(off, dtype=torch.float32, device=dev)
        # operand copies: `opc` in the model's own op dtype (scoring); an fp16 model trains on a bf16 copy (the
        # backward pass keeps gradient-sized values in the operand type, which underflow in fp16)
        # the split form scores from a flat fp16 buffer with every table / filter bank / highway matrix in its split
        # layout at three times its master offset (refreshed from the master after each update) and trains on bf16 too
        half = self.precision in ("fp16", _lib.SPLIT_PRECISION)
        opc = master.to(self._tdtype) if self.precision not in ("fp32", _lib.SPLIT_PRECISION) else None
        opc_train = master.to(torch.bfloat16) if half else opc
        spl = torch.zeros(3 * off, dtype=torch.float16, device=dev) if self.split else None
        stats = {t: (torch.from_numpy(m.copy()).to(dev), torch.from_numpy(v.copy()).to(dev)) for t, (m, v) in moving.items()}
        fold = {t: (torch.zeros(F, dtype=torch.float32, device=dev), torch.zeros(F, dtype=torch.float32, device=dev))
                for t in moving}
        pc, gc = _lib.DiscParams(), _lib.DiscParams()
        index, col_off = {}, np.concatenate([[0], np.cumsum(self.num_filters)]).astype(int)

        def build(opbuf, opcode, split=False):
            c = _lib.Discriminator.from_buffer_copy(self._c)
            c.op_dtype = opcode
            c.gemm_split = 1 if split else 0
            for key, tf_name, kind, o, a in chunks:
                mptr = master.data_ptr() + 4 * o
                optr = (opbuf.data_ptr() + 2 * o * (3 if split else 1)) if opbuf is not None else mptr
                if key[0] in ("tgt", "src"):
                    ct = getattr(c, key[0])
                    if key[1] == "emb":
                        ct.emb = optr
                    elif key[1] == "conv_w":
                        ct.conv_w[self.filter_sizes.index(key[2])] = optr
                    elif key[1] in ("hw_out", "hw_tr"):
                        setattr(ct, key[1], optr)
                elif key[0] == "head_w":
                    c.head_w = mptr
                else:
                    c.head_b = mptr
            for t in ("tgt", "src"):
                ct = getattr(c, t)
                for i in range(len(self.filter_sizes)):
                    ct.conv_scale[i] = fold[t][0].data_ptr() + 4 * int(col_off[i])
                    ct.conv_shift[i] = fold[t][1].data_ptr() + 4 * int(col_off[i])
            return c

        for key, tf_name, kind, o, a in chunks:
            mptr, gptr = master.data_ptr() + 4 * o, grad.data_ptr() + 4 * o
            index[key] = (tf_name, kind, o, a.size)
            if key[0] in ("tgt", "src"):
                pt, gt = getattr(pc, key[0]), getattr(gc, key[0])
                if key[1] == "emb":
                    pt.emb, gt.emb = mptr, gptr
                elif key[1] == "conv_w":
                    i = self.filter_sizes.index(key[2])
                    pt.conv_w[i], gt.conv_w[i] = mptr, gptr
                else:
                    setattr(pt, key[1], mptr), setattr(gt, key[1], gptr)
            elif key[0] == "head_w":
                pc.head_w, gc.head_w = mptr, gptr
            else:
                pc.head_b, gc.head_b = mptr, gptr
        for t in ("tgt", "src"):
            pt = getattr(pc, t)
            pt.moving_mean, pt.moving_var = stats[t][0].data_ptr(), stats[t][1].data_ptr()
        c = build(spl, self._op, split=True) if self.split else build(opc, self._op)
        c_train = build(opc_train, _lib.NMTGAN_BF16) if half else c
        self._keep = [master, grad, ms, stats, fold] + [t for t in (opc, opc_train, spl) if t is not None]
        self._c = c
        self._train = dict(master=master, grad=grad, ms=ms, op=opc, op_train=opc_train, c_train=c_train,
                           op_train_code=_lib.NMTGAN_BF16 if half else self._op, split=spl,
                           pc=pc, gc=gc, index=index, n=off, stats=stats, fold=fold,
                           steps=0, loss=torch.zeros(1, dtype=torch.float32, device=dev),
                           acc=torch.zeros(1, dtype=torch.float32, device=dev))
        self._fold_bn()
        self._refresh_split()
        return self._train

    def _refresh_split(self):
        """Split form: rebuild the scoring copy (tables [hi | lo | hi], filters per tap and highway matrices
        [hi | hi | lo]) from the fp32 master."""
        tr = self._train
        if tr is None or tr.get("split") is None:
            return
        master, spl, E, Fp = tr["master"], tr["split"], self.dim_word, self.Fp
        for key, (tf_name, kind, o, n) in tr["index"].items():
            src = master[o:o + n]
            if kind == "tab":
                img = _lib.split3_acts(src.view(-1, E))
            elif kind == "conv":
                w = key[2]
                img = _lib.split3_conv_weights(src.view(-1, w * E), w)
            elif kind == "hw":
                img = _lib.split3_weights(src.view(Fp, Fp))
            else:
                continue
            spl[3 * o:3 * o + 3 * n].copy_(img.reshape(-1))

    def _fold_bn(self):
        tr, F = self._train, self.num_filters_total
        for t in ("tgt", "src"):
            pt = getattr(tr["pc"], t)
            check(lib.nmtgan_disc_fold_bn(pt.conv_b, pt.gamma, pt.beta, pt.moving_mean, pt.moving_var, F,
                                          ptr(tr["fold"][t][0]), ptr(tr["fold"][t][1]), _lib.stream_ptr()), "nmtgan_disc_fold_bn")

    def train_step(self, x, y, xs, *, return_grads=False):
        """One iteration of DisCNN.train (cnn_discriminator.py:664-691) on one batch: x [max_len, B] target ids,
        y [2, B] one-hot labels, xs [max_len, B] source ids (the arrays dis_three_length_prepare feeds), dropout keep
        1.0 as train() feeds it.  Forward in training mode (BatchNorm batch moments, moving statistics moved),
        gradients of sum(losses), tower average + clip to [-1, 1] by value (share_function.py:413-441),
        the optimizer of the constructor (:153-166: Adam(), Adadelta(1., 0.95, 1e-6), SGD(1e-4), RMSProp(1e-4)), and after the FIRST step every trainable variable clipped to [-1, 1] (:690-691).
        With torch.distributed initialised each rank is one tower: the batch is cut evenly (numpy.split, :654-656),
        gradients are averaged with one all-reduce, loss and accuracy are tower means; every rank keeps its own
        BatchNorm moving statistics (the reference's towers race on the shared ones).  Returns (loss, accuracy)."""
        if self.optimizer_name not in ('rmsprop', 'adadelta', 'adam', 'sgd'):
            raise ValueError("optimizer must be adam, adadelta or sgd.")          # cnn_discriminator.py:166
        x = np.ascontiguousarray(np.asarray(x).T, dtype=np.int32)            # tf.transpose, :260-262
        xs = np.ascontiguousarray(np.asarray(xs).T, dtype=np.int32)
        lab = np.ascontiguousarray(np.asarray(y, dtype=np.float32).T)
        if x.ndim != 2 or x.shape[1] != self.max_len or xs.shape != x.shape or lab.shape != (x.shape[0], 2):
            raise ValueError("train_step: x, xs [max_len=%d, B] and y [2, B] expected" % self.max_len)
        world, rank = 1, 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            world, rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
        if x.shape[0] % world != 0 or x.shape[0] < world:
            raise ValueError("train_step: batch of %d cannot be split evenly over %d towers" % (x.shape[0], world))
        per = x.shape[0] // world
        x, xs, lab = x[rank * per:(rank + 1) * per], xs[rank * per:(rank + 1) * per], lab[rank * per:(rank + 1) * per]
        tr = self._enable_training()
        dev, M = self.device, x.shape[0]
        xd, xsd = torch.from_numpy(x).to(dev), torch.from_numpy(xs).to(dev)
        ld = torch.from_numpy(lab).to(dev)
        ct = tr["c_train"]
        need = lib.nmtgan_disc_train_workspace_bytes(C.byref(ct), M)
        if need == 0:
            raise _lib.NmtGanError("disc train workspace: " + lib.nmtgan_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        st = _lib.stream_ptr()
        check(lib.nmtgan_disc_train_grads(C.byref(ct), C.byref(tr["pc"]), C.byref(tr["gc"]), ptr(xd), ptr(xsd), ptr(ld), M,
                                          ptr(tr["loss"]), ptr(tr["acc"]), ptr(self._ws), self._ws.numel(), st),
              "nmtgan_disc_train_grads")
        la = torch.cat([tr["loss"], tr["acc"]])
        if world > 1:
            work_g, work_l = tr["grad"], la
            if torch.distributed.get_backend() != "nccl":
                work_g, work_l = tr["grad"].cpu(), la.cpu()
            torch.distributed.all_reduce(work_g)
            torch.distributed.all_reduce(work_l)
            if work_g is not tr["grad"]:
                tr["grad"].copy_(work_g)
            la = work_l.to(dev) / world
        grads = self._unpack(tr["grad"] / world) if return_grads else None
        if self.optimizer_name == 'adadelta':     # AdadeltaOptimizer(learning_rate=1., rho=0.95, epsilon=1e-6), :156-158
            if "accum" not in tr:
                tr["accum"], tr["accum_update"] = torch.zeros_like(tr["master"]), torch.zeros_like(tr["master"])
            check(lib.nmtgan_adadelta_update(ptr(tr["master"]), ptr(tr["accum"]), ptr(tr["accum_update"]), ptr(tr["grad"]),
                                             tr["n"], 1.0, 0.95, 1e-6, 1.0 / world, 1.0, ptr(tr["op_train"]), tr["op_train_code"], st),
                  "nmtgan_adadelta_update")
        elif self.optimizer_name == 'adam':       # tf.train.AdamOptimizer() defaults, :153-155
            if "adam_m" not in tr:
                tr["adam_m"], tr["adam_v"] = torch.zeros_like(tr["master"]), torch.zeros_like(tr["master"])
            check(lib.nmtgan_adam_update(ptr(tr["master"]), ptr(tr["adam_m"]), ptr(tr["adam_v"]), ptr(tr["grad"]), tr["n"],
                                         0.001, 0.9, 0.999, 1e-8, tr["steps"] + 1, 1.0 / world, 1.0, ptr(tr["op_train"]),
                                         tr["op_train_code"], st),
                  "nmtgan_adam_update")
        elif self.optimizer_name == 'sgd':        # GradientDescentOptimizer(0.0001), :159-161
            check(lib.nmtgan_sgd_update(ptr(tr["master"]), ptr(tr["grad"]), tr["n"], self.learning_rate, 1.0 / world, 1.0,
                                        ptr(tr["op_train"]), tr["op_train_code"], st), "nmtgan_sgd_update")
        else:
            check(lib.nmtgan_rmsprop_update(ptr(tr["master"]), ptr(tr["ms"]), None, ptr(tr["grad"]), tr["n"],
                                            self.learning_rate, 0.9, 0.0, 1e-10, 1.0 / world, 1.0, ptr(tr["op_train"]),
                                            tr["op_train_code"], st), "nmtgan_rmsprop_update")
        tr["steps"] += 1
        if tr["steps"] == 1:
            check(lib.nmtgan_clip_values(ptr(tr["master"]), tr["n"], -1.0, 1.0, ptr(tr["op_train"]), tr["op_train_code"], st),
                  "nmtgan_clip_values")
        if tr["op"] is not None and tr["op"] is not tr["op_train"]:        # fp16 scoring copy from the updated master
            check(lib.nmtgan_clip_values(ptr(tr["master"]), tr["n"], float("-inf"), float("inf"), ptr(tr["op"]), self._op, st),
                  "nmtgan_clip_values")
        self._fold_bn()
        self._refresh_split()
        torch.cuda.synchronize(dev)
        out = (float(la[0].item()), float(la[1].item()))
        return out + (grads,) if return_grads else out

    def _unpack(self, flat):
        """Flat packed buffer -> {TF variable name: array in the variable's TF shape} (trainable variables)."""
        tr, E, F, Fp = self._train, self.dim_word, self.num_filters_total, self.Fp
        host = flat.cpu().numpy()
        out = {}
        col_off = np.concatenate([[0], np.cumsum(self.num_filters)]).astype(int)
        for key, (tf_name, kind, o, n) in tr["index"].items():
            a = host[o:o + n]
            if kind == "tab":
                out[tf_name] = a.reshape(-1, E).copy()
            elif kind == "conv":
                w = key[2]
                nf = self.num_filters[self.filter_sizes.index(w)]
                out[tf_name] = np.ascontiguousarray(a.reshape(nf, w * E).T).reshape(w, E, 1, nf)
            elif kind == "vec":
                for i, w in enumerate(self.filter_sizes):
                    out[tf_name % w] = a[col_off[i]:col_off[i + 1]].copy()
            elif kind == "hw":
                out[tf_name] = np.ascontiguousarray(a.reshape(Fp, Fp)[:F, :F].T)
            elif kind == "head":
                h = a.reshape(2, 2 * Fp)
                out[tf_name] = np.ascontiguousarray(np.concatenate([h[:, :F], h[:, Fp:Fp + F]], axis=1).T)
            else:
                out[tf_name] = a.copy()
        return out

    def get_weights(self) -> Dict[str, np.ndarray]:
        """Every variable of the scope under its TF name and shape (what tf.train.Saver(params in scope) writes,
        cnn_discriminator.py:202-205): trainable variables + BatchNorm moving statistics."""
        if self._train is None:
            return {k: np.asarray(self._W_src[k], np.float32) for k in self._all_names()}
        out = self._unpack(self._train["master"])
        col_off = np.concatenate([[0], np.cumsum(self.num_filters)]).astype(int)
        for tkey, _, conv_fmt, _, _ in self._tf_names():
            mm, mv = (t.cpu().numpy() for t in self._train["stats"][tkey])
            for i, w in enumerate(self.filter_sizes):
                bn = conv_fmt % w + "conv_batch_norm/bn/"
                out[bn + "moving_mean"] = mm[col_off[i]:col_off[i + 1]].copy()
                out[bn + "moving_variance"] = mv[col_off[i]:col_off[i + 1]].copy()
        return out

    def _all_names(self):
        names = [self.scope + "/model_self/W", self.scope + "/model_self/b"]
        for _, emb_name, conv_fmt, out_name, tr_name in self._tf_names():
            names += [emb_name, out_name, tr_name]
            for w in self.filter_sizes:
                p = conv_fmt % w
                names += [p + "self_model/W", p + "self_model/b"] + [p + "conv_batch_norm/bn/" + k for k in
                                                                      ("gamma", "beta", "moving_mean", "moving_variance")]
        return names

    def save(self, prefix=None):
        """saver.save(sess, saveto) (cnn_discriminator.py:699, gan_train.py:161): a TF V2 bundle of the scope."""
        from . import tf_checkpoint
        tf_checkpoint.save_checkpoint(str(self.saveto if prefix is None else prefix), self.get_weights(), update_state=False)

    @property
    def c_struct(self):
        if self._c is None:
            raise RuntimeError("DisCNN: load_weights() has not been called")
        return self._c

    def _workspace(self, chunk):
        need = lib.nmtgan_disc_workspace_bytes(C.byref(self.c_struct), chunk)
        if need == 0:
            raise _lib.NmtGanError("disc workspace: " + lib.nmtgan_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    # ------------------------------------------------------------------ inference
    def features(self, ids_BT, source: bool) -> torch.Tensor:
        """Highway features [M, Fp] of one tower for batch-major ids [M, T]."""
        ids = torch.as_tensor(np.ascontiguousarray(ids_BT), dtype=torch.int32).to(self.device).contiguous()
        M, T = ids.shape
        if T != self.max_len:
            raise ValueError("ids must have max_len=%d columns, got %d" % (self.max_len, T))
        chunk = max(1, min(self.chunk_rows, M))
        ws = self._workspace(chunk)
        feat = torch.empty((M, self.Fp), dtype=torch.float32, device=self.device)
        check(lib.nmtgan_disc_features(C.byref(self.c_struct), 1 if source else 0, ptr(ids), M, ptr(feat), chunk,
                                       ptr(ws), ws.numel(), _lib.stream_ptr()), "nmtgan_disc_features")
        return feat

    def run_ypred_for_auc(self, dis_input_x, dis_input_xs, dis_dropout_keep_prob=1.0) -> np.ndarray:
        """sess.run(disc.dis_ypred_for_auc, {dis_input_x:[T,B], dis_input_xs:[T,B], keep_prob:1.0})
        (model.py:311-313): softmax probabilities [B, 2] (column 1 = P(real))."""
        if float(dis_dropout_keep_prob) != 1.0:
            raise ValueError("inference graph: dis_dropout_keep_prob must be 1.0")
        x = np.ascontiguousarray(np.asarray(dis_input_x).T)        # tf.transpose, cnn_discriminator.py:537-538
        xs = np.ascontiguousarray(np.asarray(dis_input_xs).T)
        if x.shape != xs.shape:
            raise ValueError("dis_input_x and dis_input_xs must have the same [max_len, batch] shape")
        ft = self.features(x, source=False)
        fs = self.features(xs, source=True)
        M = x.shape[0]
        probs = torch.empty((M, 2), dtype=torch.float32, device=self.device)
        check(lib.nmtgan_disc_head(C.byref(self.c_struct), ptr(ft), ptr(fs), None, M, None, ptr(probs),
                                   _lib.stream_ptr()), "nmtgan_disc_head")
        torch.cuda.synchronize(self.device)
        return probs.cpu().numpy()

    # ------------------------------------------------------------------ out of scope
    def build_train_model(self, *a, **k):
        """Reference: cnn_discriminator.py:561-591 (graph construction; nothing to build here - see train_step)."""

    def train(self, max_epoch=None, positive_data=None, negative_data=None, source_data=None):
        raise NotImplementedError("DisCNN.train reads its batches with disThreeTextIterator / shuffle.py (data pipeline, out of "
                                  "scope): feed each batch to train_step(x, y, xs) instead")
