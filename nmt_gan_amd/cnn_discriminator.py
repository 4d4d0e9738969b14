"""DisCNN facade: the inference surface of the reference's cnn_discriminator.DisCNN that
Model.get_reward drives (model.py:311-313), backed by libnmtgan_hip.so.

Only inference is in scope: the two-tower multi-width CNN + highway + 2-way
softmax of build_discriminator_body (cnn_discriminator.py:440-522) with BatchNorm in
moving-statistics mode.  Training (build_train_model / train) is out of scope and raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr

BN_EPS = 1e-3          # tf.contrib.layers.batch_norm default epsilon (cnn_discriminator.py:22-33)


def _op_torch_dtype(precision: str):
    if precision == "bf16":
        return torch.bfloat16, _lib.NMTGAN_BF16
    if precision == "fp32":
        return torch.float32, _lib.NMTGAN_F32
    raise ValueError("precision must be 'bf16' or 'fp32'")


class DisCNN(object):
    """Same constructor signature as the reference (cnn_discriminator.py:117-119); the data-path,
    optimizer and training arguments are accepted and ignored.  `precision`/`device` are
    build-only keyword extensions."""

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
                 scope='discnn', init_device="/cpu:0", reuse_var=False, precision="bf16", device="cuda:0",
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
        self.precision = precision
        self.device = torch.device(device)
        self.chunk_rows = int(chunk_rows)
        self._tdtype, self._op = _op_torch_dtype(precision)
        if len(self.filter_sizes) != len(self.num_filters) or len(self.filter_sizes) > _lib.MAX_WIDTHS:
            raise ValueError("filter_sizes/num_filters mismatch or more than %d widths" % _lib.MAX_WIDTHS)
        if max(self.filter_sizes) > self.max_len:
            raise ValueError("filter width exceeds max_len")
        self.Fp = (self.num_filters_total + 63) // 64 * 64
        self._keep = []            # device tensors kept alive for the C struct
        self._c = None
        self._ws = None
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

        c = _lib.Discriminator()
        c.op_dtype = self._op
        c.max_len, c.dim_word, c.vocab, c.vocab_s = self.max_len, E, self.vocab_size, self.vocab_size_s
        c.n_widths = len(self.filter_sizes)
        for i, (w, n) in enumerate(zip(self.filter_sizes, self.num_filters)):
            c.widths[i], c.n_filters[i] = w, n
        c.F, c.Fp = F, Fp

        def tower(ct, emb_name, conv_fmt, out_name, tr_name):
            emb = np.asarray(W[s + emb_name], np.float32)
            ct.emb = ptr(dv(emb, td))
            for i, (w, n) in enumerate(zip(self.filter_sizes, self.num_filters)):
                p = s + conv_fmt % w + "/"
                Wc = np.asarray(W[p + "self_model/W"], np.float32).reshape(w * E, n)
                b = np.asarray(W[p + "self_model/b"], np.float32)
                bn = p + "conv_batch_norm/bn/"
                gamma, beta = np.asarray(W[bn + "gamma"], np.float32), np.asarray(W[bn + "beta"], np.float32)
                mean, var = np.asarray(W[bn + "moving_mean"], np.float32), np.asarray(W[bn + "moving_variance"], np.float32)
                inv = (np.float32(1.0) / np.sqrt(var + np.float32(BN_EPS))).astype(np.float32) * gamma
                shift = (b * inv + (beta - mean * inv)).astype(np.float32)
                ct.conv_w[i] = ptr(dv(Wc.T, td))
                ct.conv_scale[i] = ptr(dv(inv, torch.float32))
                ct.conv_shift[i] = ptr(dv(shift, torch.float32))
            for field, name in (("hw_out", out_name), ("hw_tr", tr_name)):
                Wh = np.zeros((Fp, Fp), np.float32)
                Wh[:F, :F] = np.asarray(W[s + name + "/weights"], np.float32).T       # Wt[out][in]
                setattr(ct, field, ptr(dv(Wh, td)))

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
        return self

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
        raise NotImplementedError("discriminator training is outside the rollout-reward path (SURVEY section 8f-4)")

    train = build_train_model
