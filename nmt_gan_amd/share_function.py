"""Host-side numpy helpers on the rollout-reward path (same names as the reference's
share_function.py) plus the rollout plan builder used by Model.get_reward."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

PAD_ID, UNK_ID, BOS_ID, EOS_ID = 0, 1, 2, 3      # reference utils.py:39, 269, 287


def _first_eos_cut(y_sample, dicts, width):
    """Rows of `y_sample` (a 2-D array or a list of id sequences of any lengths) cut before their first </S>
    and laid into [n, width]; the id of </S> is `dicts.get('</S>')` exactly as in the reference - a map
    without that key therefore cuts nothing - and id 3 when no map is given (keyword-free extension)."""
    eos = EOS_ID if dicts is None else dicts.get("</S>")
    n = len(y_sample)
    y = np.zeros((n, width), dtype=np.int32)
    mask = np.zeros((n, width), dtype=np.float32)
    if isinstance(y_sample, np.ndarray) and y_sample.ndim == 2 and y_sample.shape[1] <= width:
        L = y_sample.shape[1]
        hit = (y_sample == eos) if eos is not None else np.zeros(y_sample.shape, bool)
        first = np.where(hit.any(axis=1), hit.argmax(axis=1), L)
        keep = np.arange(L)[None, :] < first[:, None]
        y[:, :L] = np.where(keep, y_sample, 0)
        mask[:, :L] = keep
        return y, mask
    for i, s in enumerate(y_sample):
        s = np.asarray(s).reshape(-1)
        hit = np.flatnonzero(s == eos) if eos is not None else ()
        first = int(hit[0]) if len(hit) else len(s)
        if first > width:
            raise ValueError("sentence %d has %d tokens before </S>, more than %d columns" % (i, first, width))
        y[i, :first] = s[:first]
        mask[i, :first] = 1.0
    return y, mask


def deal_generated_samples(y_sample, dicts=None):
    """Zero every position from the first </S> on; mask = 1 before it (</S> not included).
    Mirrors share_function.py:269-289; the output is as wide as the longest input row."""
    width = y_sample.shape[1] if isinstance(y_sample, np.ndarray) and y_sample.ndim == 2 else \
        max([len(s) for s in y_sample] or [0])
    return _first_eos_cut(y_sample, dicts, width)


def deal_generated_samples_to_maxlen(y_sample, dicts, maxlen):
    """Same cut, laid into [n, maxlen] (share_function.py:291-309; gan_train.py:123 prepares the ground-truth
    batch of the teacher-forcing update with it).  Like the reference's slice assignment it raises when a
    sentence keeps more than maxlen tokens."""
    return _first_eos_cut(y_sample, dicts, int(maxlen))


def extend_sentence_to_maxlen(seqs, maxlen=50):
    """Zero-pad every sentence to maxlen (share_function.py:240-245); like the reference it
    raises if a sentence is longer than maxlen."""
    x = np.zeros((len(seqs), maxlen), dtype=np.int32)
    for i, s in enumerate(seqs):
        s = np.asarray(s)
        if len(s) > maxlen:
            raise ValueError("sentence %d has length %d > maxlen %d" % (i, len(s), maxlen))
        x[i, :len(s)] = s
    return x


class FlushFile(object):
    """stdout wrapper that flushes after every write (share_function.py:75-84; gan_train.py:167 installs it)."""

    def __init__(self, f):
        self.f = f

    def write(self, x):
        self.f.write(x)
        self.f.flush()

    def flush(self):
        self.f.flush()


def prepare_gan_dis_data(train_data_source, train_data_target, gan_dis_source_data, gan_dis_positive_data,
                         num=None, reshuf=True, seed=None):
    """Reference: share_function.py:17-36 (+ shuffle.py): draw `num` aligned (source, target) line pairs of the training
    corpus as the discriminator's source / positive data for this GAN iteration (gan_train.py:137-144).  The
    reference shells out (`python shuffle.py`, `head -n`, `rm`) through os.popen without waiting for the commands; here
    the same files are produced in-process: lines joined pairwise, shuffled (random.shuffle; `seed` is a keyword-only
    style extension for reproducible runs), the first `num` pairs written.  With reshuf=False the reference reads
    existing `<file>.shuf` copies; they are used when present, else the corpus order is kept.  The `.shuf`
    intermediates are removed as the reference does.  Returns num."""
    import os
    import random

    def read(path):
        with open(path, 'r') as f:
            return [l.rstrip('\n') for l in f]

    src_path, dst_path = train_data_source, train_data_target
    if not reshuf and os.path.exists(src_path + '.shuf') and os.path.exists(dst_path + '.shuf'):
        src_path, dst_path = src_path + '.shuf', dst_path + '.shuf'
    src, dst = read(src_path), read(dst_path)
    if len(src) != len(dst):
        raise ValueError("prepare_gan_dis_data: %s and %s have different line counts" % (src_path, dst_path))
    total = len(read(train_data_source))
    if num is None or num > total:
        num = total
    pairs = list(zip(src, dst))
    if reshuf:
        (random.Random(seed) if seed is not None else random).shuffle(pairs)
    with open(gan_dis_source_data, 'w') as fs, open(gan_dis_positive_data, 'w') as fp:
        for a, b in pairs[:num]:
            fs.write(a + '\n')
            fp.write(b + '\n')
    for path in (train_data_source + '.shuf', train_data_target + '.shuf'):
        if os.path.exists(path):
            os.remove(path)
    return num


def timing_signal(length, channels):
    """Sinusoid table of add_timing_signal_1d (tensor2tensor/common_attention.py:55-67), evaluated
    in fp32 like the TF graph: sines in the first half of the channels, cosines in the second."""
    position = np.arange(length, dtype=np.float32)
    nt = channels // 2
    inc = np.float32(math.log(1.0e4)) / (np.float32(nt) - np.float32(1.0))
    inv = np.exp(np.arange(nt, dtype=np.float32) * -inc).astype(np.float32)
    scaled = position[:, None] * inv[None, :]
    sig = np.concatenate([np.sin(scaled), np.cos(scaled)], axis=1).astype(np.float32)
    if channels % 2:
        sig = np.pad(sig, [[0, 0], [0, 1]])
    return sig


def shift_right(y, pad=BOS_ID):
    """Decoder input [<S>, y_0 .. y_{T-2}] (model.py:764-766)."""
    y = np.asarray(y)
    return np.concatenate([np.full_like(y[:, :1], pad), y[:, :-1]], axis=1)


def roll_shard_pieces(B, rollnum, world, rank):
    """Roll-level spill of the (sentence, roll) sharding for B < world: the B*rollnum (b, r) units, sentence-major,
    are cut with the split_tensor boundaries; returns (u0, u1, [(b, r0, r1), ..]) - this rank's unit range and the
    (sentence, roll range) pieces inside it.  All prefixes g of a (b, r) stay on one rank."""
    bounds = split_bounds(B * rollnum, world)
    u0, u1 = bounds[rank], bounds[rank + 1]
    pieces = []
    b = u0 // rollnum if rollnum else 0
    while b * rollnum < u1 and u0 < u1:
        r0, r1 = max(u0, b * rollnum) - b * rollnum, min(u1, (b + 1) * rollnum) - b * rollnum
        if r1 > r0:
            pieces.append((b, r0, r1))
        b += 1
    return u0, u1, pieces


def split_bounds(batch_size, n):
    """Shard boundaries of split_tensor (model.py:742-753): int(linspace(0, B, n+1))."""
    return [int(v) for v in np.linspace(0.0, float(batch_size), n + 1).astype(np.float32).astype(np.int32)]


@dataclass
class RolloutPlan:
    """Row order of the lock-step rollout: one row per live (sentence b, roll r, prefix g), sorted by
    remaining private steps (descending) so that the live rows of step s are the prefix [0, live[s])."""
    B: int
    T: int
    rollnum: int
    lens: np.ndarray          # [B] mask length of each sentence
    row_b: np.ndarray         # [n_rows] int32 (local sentence index)
    row_r: np.ndarray
    row_g: np.ndarray
    row_lrow: np.ndarray      # b*T + g
    row_uoff: np.ndarray      # int64 offset of u[r, g-1, b_global, g]
    grp_first: np.ndarray     # [B, T-1] int32 first row of (b, g) or -1
    live: np.ndarray          # [n_steps+1] int32
    step_off: np.ndarray      # [n_steps+1] int64
    grp_slots: int = 0        # groups of one sentence served by one attention workgroup (32 // rollnum)
    grp_table: np.ndarray = None   # [n_entries, 2 + 2*grp_slots] int32: blk, 0, row0[slots], cnt[slots]
    grp_live: np.ndarray = None    # [n_steps+1] int32 entries whose first slot is live at step s

    @property
    def n_rows(self):
        return int(self.row_b.shape[0])

    @property
    def n_steps(self):
        return int(self.live.shape[0] - 1)


def build_rollout_plan(y_sample_mask, rollnum, b_offset=0, B_global=None, r_offset=0) -> RolloutPlan:
    """Rows (b, r, g) for 1 <= g < len_b: for g >= len_b the masked rollout equals y_sample[b]
    (model.py:307), so those are scored once through p_base.  Row (b, r, g) samples
    positions g .. len_b-1: the first from the shared prefix logits, then len_b-1-g private steps.
    Uniforms are addressed by GLOBAL sentence index (local b + b_offset) and GLOBAL roll index (local r +
    r_offset, for a shard that holds only some rolls of a sentence) so any sharding draws identical tokens."""
    mask = np.asarray(y_sample_mask)
    B, T = mask.shape
    lens = (mask != 0).sum(axis=1).astype(np.int64)
    if not np.array_equal(mask != 0, np.arange(T)[None, :] < lens[:, None]):
        raise ValueError("y_sample_mask must be a prefix of ones per row (deal_generated_samples output)")
    if B_global is None:
        B_global = B
    bs, gs = [], []
    for b in range(B):
        g = np.arange(1, int(lens[b]), dtype=np.int64)
        bs.append(np.full(g.shape, b, np.int64))
        gs.append(g)
    bb = np.concatenate(bs) if bs else np.zeros(0, np.int64)
    gg = np.concatenate(gs) if gs else np.zeros(0, np.int64)
    steps = lens[bb] - 1 - gg
    # primary: steps desc, then g, then b - rows of one give_num are adjacent inside a class of equal remaining
    # steps, so the discriminator's partial-window reuse can cut its sub-ranges where give_num // taps changes
    order = np.lexsort((bb, gg, -steps))
    bb, gg, steps = bb[order], gg[order], steps[order]
    n_grp = bb.shape[0]
    row_b = np.repeat(bb, rollnum)
    row_g = np.repeat(gg, rollnum)
    row_r = np.tile(np.arange(rollnum, dtype=np.int64), n_grp)
    row_steps = np.repeat(steps, rollnum)
    grp_first = np.full((B, max(T - 1, 1)), -1, np.int32)
    grp_first[bb, gg - 1] = (np.arange(n_grp) * rollnum).astype(np.int32)
    n_steps = int(steps.max()) if n_grp else 0
    live = np.array([(row_steps >= s).sum() for s in range(n_steps + 1)], dtype=np.int32)
    live[0] = row_b.shape[0]
    step_off = np.zeros(n_steps + 1, np.int64)
    step_off[1:] = np.cumsum(live[1:].astype(np.int64))
    row_uoff = (((row_r + int(r_offset)) * (T - 1) + (row_g - 1)) * B_global + (row_b + b_offset)) * T + row_g
    slots, table, tlive = group_slot_table(grp_first, lens, rollnum, live)
    return RolloutPlan(grp_slots=slots, grp_table=table, grp_live=tlive,
                       B=B, T=T, rollnum=rollnum, lens=lens, row_b=row_b.astype(np.int32),
                       row_r=row_r.astype(np.int32), row_g=row_g.astype(np.int32),
                       row_lrow=(row_b * T + row_g).astype(np.int32), row_uoff=row_uoff.astype(np.int64),
                       grp_first=grp_first, live=live, step_off=step_off)


def group_slot_table(grp_first, lens, rollnum, live):
    """Slot table of nmtgan_attn_group (include/nmtgan.h): consecutive prefixes g, g+1, .. of ONE sentence
    share their keys (encoder memory; teacher-forced prefix up to the largest g), so 32 // rollnum of them
    ride side by side in the 32 MFMA columns of one workgroup.  Entries are sorted by the row index of their
    first slot (the smallest g = most remaining steps), so the entries live at step s are a prefix."""
    slots = max(1, 32 // int(rollnum)) if rollnum <= 32 else 0
    n_steps = live.shape[0] - 1
    if slots == 0 or grp_first.size == 0:
        return 0, np.zeros((0, 2), np.int32), np.zeros(n_steps + 1, np.int32)
    ents = []
    for b in range(grp_first.shape[0]):
        gs = [g for g in range(1, int(lens[b])) if grp_first[b, g - 1] >= 0]
        for i in range(0, len(gs), slots):
            part = gs[i:i + slots]
            rows = [int(grp_first[b, g - 1]) for g in part] + [-1] * (slots - len(part))
            cnts = [g + 1 for g in part] + [0] * (slots - len(part))
            ents.append([b, 0] + rows + cnts)
    if not ents:
        return slots, np.zeros((0, 2 + 2 * slots), np.int32), np.zeros(n_steps + 1, np.int32)
    table = np.asarray(ents, np.int32)
    table = table[np.argsort(table[:, 2], kind="stable")]
    tlive = np.array([(table[:, 2] < live[s]).sum() for s in range(n_steps + 1)], np.int32)
    return slots, np.ascontiguousarray(table), tlive
