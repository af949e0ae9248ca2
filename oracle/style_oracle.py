"""ORACLE — test infrastructure only, never shipped on the product path.

A functional fp32 torch-CPU restatement of the reference's hot path
(StyleTransferModel.forward -> get_total_loss -> backward -> Adam), written
against a flat parameter dict keyed by the reference's state_dict names.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Pinned: tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which were produced by running the reference itself
(tests/golden/make_golden.py).  Backward is torch autograd over this forward.

Every function cites the reference lines it restates (paths relative to
/root/reference).
"""
import math

import torch
import torch.nn.functional as F

NF, NPF, NUF, NOCT, NDEG, NUN, NMODES = 10, 5, 2, 8, 7, 47, 2   # style/model.py:13-20
NPN = NOCT * NDEG
BPM_MIN, BPM_RANGE = 50, 150                                     # style/model.py:22-25
EPS = 1e-7                                                       # style/model.py:11


def mean_size(*values, factor=1):
    """style/model.py:31-33."""
    return math.ceil(sum(values) / len(values) * factor)


def lrelu(x):
    return F.leaky_relu(x)          # default slope 0.01 everywhere in the reference


class Params:
    """Prefix view over a flat {state_dict name: tensor} mapping."""

    def __init__(self, flat, prefix=''):
        self.flat, self.prefix = flat, prefix

    def sub(self, name):
        return Params(self.flat, self.prefix + name + '.')

    def __getitem__(self, name):
        return self.flat[self.prefix + name]

    def lin(self, name, x, act=False):
        y = F.linear(x, self[name + '.weight'], self[name + '.bias'])
        return lrelu(y) if act else y


# ---------------------------------------------------------------- LSTM (style/utils/pytorch.py:19-25)
def lstm_dir(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """One direction, batch-first, zero initial state; torch gate order i,f,g,o.
    x: (batch, steps, in) -> (batch, steps, hidden)."""
    B, S, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    zx = F.linear(x, w_ih, b_ih)
    outs = [None] * S
    order = range(S - 1, -1, -1) if reverse else range(S)
    for s in order:
        z = zx[:, s] + F.linear(h, w_hh, b_hh)
        i, f, g, o = z.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs[s] = h
    return torch.stack(outs, 1)


def lstm(P, name, x, bidirectional=False, fast=False):
    """x: (batch, steps, in). Output (batch, steps, H * dirs)."""
    names = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']
    w = [P[f'{name}.{n}'] for n in names]
    if bidirectional:
        w += [P[f'{name}.{n}_reverse'] for n in names]
    if fast:        # same op the reference's nn.LSTM dispatches to; used for the cpu_baseline timing
        H = w[1].shape[1]
        D = 2 if bidirectional else 1
        z = x.new_zeros(D, x.shape[0], H)
        return torch._VF.lstm(x, (z, z), w, True, 1, 0.0, False, bidirectional, True)[0]
    y = lstm_dir(x, *w[:4])
    if bidirectional:
        y = torch.cat([y, lstm_dir(x, *w[4:], reverse=True)], -1)
    return y


# ---------------------------------------------------------------- combine (style/model.py:796-815)
def combine(x, dim):
    """Norm-weighted merge over `dim`: n_c = sqrt(1 + sum of squares of slice c)."""
    other = [d for d in range(x.dim()) if d != dim]
    n = torch.sqrt(1.0 + (x * x).sum(other, keepdim=True))
    return (x * n).sum(dim) / n.sum()


def combine_pair(a, b):
    return combine(torch.stack([a, b]), 0)


# ---------------------------------------------------------------- encoders
def pitched_channels_encoder(P, x, instr, fast=False):
    """style/model.py:77-99. x (1,C,R,T,10,56,5), instr (1,C,I) -> beats (1,C,R,T,H), bars (1,R,2*Hb)."""
    B, C, R, T = x.shape[:4]
    v = x.transpose(-1, -2).reshape(B * C * R * T, NF * NPF, NPN)          # channel = f*5+feat
    v = F.conv1d(v, P['beats_conv.module.weight'], P['beats_conv.module.bias'],
                 stride=NDEG, padding=4)
    x1 = lrelu(v).reshape(B, C, R, T, -1)                                   # idx = ch*8 + octave
    x2 = P.lin('instruments_linear', instr, act=True)[:, :, None, None, :].expand(B, C, R, T, -1)
    v = P.lin('linear', torch.cat([x1, x2], -1), act=True)
    H = v.shape[-1]
    beats = lstm(P, 'beats_lstm.module', v.reshape(B * C * R, T, H), fast=fast).reshape(B, C, R, T, -1)
    last = combine(beats[:, :, :, -1], 1)
    bars = lstm(P, 'bars_lstm', last, bidirectional=True, fast=fast)
    return beats, bars


def unpitched_channels_encoder(P, x, fast=False):
    """style/model.py:128-141. x (1,1,R,T,10,47,2)."""
    B, C, R, T = x.shape[:4]
    v = x.transpose(-1, -2).reshape(B, C, R, T, -1)                          # idx = f*94 + feat*47 + note
    v = P.lin('linear', v, act=True)
    beats = lstm(P, 'beats_lstm.module', v.reshape(B * C * R, T, -1), fast=fast).reshape(B, C, R, T, -1)
    last = combine(beats[:, :, :, -1], 1)
    bars = lstm(P, 'bars_lstm', last, bidirectional=True, fast=fast)
    return beats, bars


def _bcast_cat(parts, shape):
    return torch.cat([p.expand(*shape, p.shape[-1]) for p in parts], -1)


def pitched_rhythm_encoder(P, beats, bars, x, instr, mode, bpm):
    """style/model.py:346-381 -> (1,R,T,10,rhythm)."""
    B, C, R, T = x.shape[:4]
    parts = [
        P.lin('beats_linear', beats, act=True)[:, :, :, :, None, :],
        P.lin('bars_linear', bars, act=True)[:, None, :, None, None, :],
        P.lin('channels_linear', x.reshape(B, C, R, T, NF, -1), act=True),
        P.lin('instruments_linear', instr, act=True)[:, :, None, None, None, :],
        P.lin('mode_linear', mode, act=True)[:, None, None, None, None, :],
        P.lin('bpm_linear', bpm[:, None], act=True)[:, None, None, None, None, :],
    ]
    v = P.lin('linear', _bcast_cat(parts, (B, C, R, T, NF)), act=True)
    return combine(v, 1)


def unpitched_rhythm_encoder(P, beats, bars, x, bpm):
    """style/model.py:418-443."""
    B, C, R, T = x.shape[:4]
    parts = [
        P.lin('beats_linear', beats, act=True)[:, :, :, :, None, :],
        P.lin('bars_linear', bars, act=True)[:, None, :, None, None, :],
        P.lin('channels_linear', x.reshape(B, C, R, T, NF, -1), act=True),
        P.lin('bpm_linear', bpm[:, None], act=True)[:, None, None, None, None, :],
    ]
    v = P.lin('linear', _bcast_cat(parts, (B, C, R, T, NF)), act=True)
    return combine(v, 1)


def style_encoder(P, bars, instr, mode, bpm, fast=False):
    """style/model.py:179-200 -> (1,style)."""
    B, C = instr.shape[:2]
    last = lstm(P, 'bars_lstm', bars, fast=fast)[:, -1]
    parts = [
        last[:, None, :],
        P.lin('instruments_linear', instr, act=True),
        P.lin('mode_linear', mode, act=True)[:, None, :],
        P.lin('bpm_linear', bpm[:, None], act=True)[:, None, :],
    ]
    v = P.lin('linear', _bcast_cat(parts, (B, C)), act=True)
    return combine(v, 1)


def _octave_degree(P, y, width):
    """The shared 'octave (+) scale degree' outer sum: style/model.py:270-286 and :644-660.
    y (..., K) -> (..., 56, width) with note = octave*7 + degree."""
    o = lrelu(P.lin('octave_linear', y).reshape(*y.shape[:-1], NOCT, 1, width))
    d = lrelu(P.lin('scale_degree_linear', y).reshape(*y.shape[:-1], 1, NDEG, width))
    return lrelu(o + d).reshape(*y.shape[:-1], NPN, width)


def melody_encoder(P, beats, bars, x, instr):
    """style/model.py:252-297 -> (1,R,T,10,56,melody). The octave/degree part has no
    beat-fraction axis (size-1 broadcast), exactly as in the reference."""
    B, C, R, T = x.shape[:4]
    parts = [
        P.lin('beats_linear', beats, act=True),
        P.lin('bars_linear', bars, act=True)[:, None, :, None, :],
        P.lin('instruments_linear', instr, act=True)[:, :, None, None, :],
    ]
    y = _bcast_cat(parts, (B, C, R, T))
    width = P['linear.weight'].shape[0]
    od = _octave_degree(P, y, width)[:, :, :, :, None]                       # (B,C,R,T,1,56,w)
    ch = P.lin('channels_linear', x, act=True)                               # (B,C,R,T,10,56,7)
    v = torch.cat([od.expand(B, C, R, T, NF, NPN, width), ch], -1)
    return combine(P.lin('linear', v, act=True), 1)


def song_info(P, style, rhythm, fast=False):
    """style/model.py:513-562 -> instruments logits (1,41), mode logits (1,2), bpm (1,)."""
    B, R, T = rhythm.shape[:3]
    v = lstm(P, 'beats_lstm.module', rhythm.reshape(B * R, T, -1), fast=fast)[:, -1].reshape(B, R, -1)
    feats = lstm(P, 'bars_lstm', v, fast=fast)[:, -1]

    def head(name):
        a = P.lin(f'style_{name}_linear', style, act=True)
        b = P.lin(f'rhythm_{name}_linear', feats, act=True)
        return P.lin(f'{name}_linear', torch.cat([a, b], -1))

    bpm = torch.sigmoid(head('bpm')[:, 0]) * BPM_RANGE + BPM_MIN
    return head('instruments'), head('mode'), bpm


def pitched_style_applier(P, style, melody, rhythm, instr):
    """style/model.py:624-675 -> (1,C,R,T,10,56,5)."""
    B, C = instr.shape[:2]
    R, T = rhythm.shape[1:3]
    parts = [
        P.lin('style_linear', style, act=True)[:, None, None, None, None, :],
        P.lin('rhythm_linear', rhythm, act=True)[:, None],
        P.lin('instruments_linear', instr, act=True)[:, :, None, None, None, :],
    ]
    y = _bcast_cat(parts, (B, C, R, T, NF))
    od = _octave_degree(P, y, NPF * 6)                                       # (B,C,R,T,10,56,30)
    mel = P.lin('melody_linear', melody, act=True)[:, None].expand(B, C, R, T, NF, NPN, -1)
    z = P.lin('linear', torch.cat([od, mel], -1))
    return torch.cat([6.0 * torch.sigmoid(z[..., :1]), torch.sigmoid(z[..., 1:])], -1)


def unpitched_style_applier(P, style, rhythm):
    """style/model.py:703-724 -> (1,1,R,T,10,47,2)."""
    B, R, T = rhythm.shape[:3]
    s = P.lin('style_linear', style, act=True).reshape(B, 1, 1, NF, -1)
    r = P.lin('rhythm_linear', rhythm, act=True)
    v = P.lin('notes_linear', _bcast_cat([s, r], (B, R, T, NF)), act=True)
    z = P.lin('linear', v.reshape(B, R, T, NF, NUN, -1))
    return torch.cat([6.0 * torch.sigmoid(z[..., :1]), torch.sigmoid(z[..., 1:])], -1)[:, None]


# ---------------------------------------------------------------- model wiring (style/model.py:751-793)
def extract_style(flat, mode, bpm, pitched, instr, unpitched=None, fast=False, mids=None):
    P = Params(flat)
    pb, pbars = pitched_channels_encoder(P.sub('pitched_channels_encoder'), pitched, instr, fast)
    prh = pitched_rhythm_encoder(P.sub('pitched_rhythm_encoder'), pb, pbars, pitched, instr, mode, bpm)
    if unpitched is None:
        bars, rhythm = pbars, prh
    else:
        ub, ubars = unpitched_channels_encoder(P.sub('unpitched_channels_encoder'), unpitched, fast)
        urh = unpitched_rhythm_encoder(P.sub('unpitched_rhythm_encoder'), ub, ubars, unpitched, bpm)
        bars, rhythm = combine_pair(pbars, ubars), combine_pair(prh, urh)
    style = style_encoder(P.sub('style_encoder'), bars, instr, mode, bpm, fast)
    melody = melody_encoder(P.sub('melody_encoder'), pb, pbars, pitched, instr)
    if mids is not None:
        mids.update(pitched_beats=pb, pitched_bars=pbars, pitched_rhythm=prh, bars=bars)
        if unpitched is not None:
            mids.update(unpitched_beats=ub, unpitched_bars=ubars, unpitched_rhythm=urh)
    return style, melody, rhythm


def forward(flat, mode, bpm, pitched, instr, unpitched=None, fast=False, mids=None):
    P = Params(flat)
    style, melody, rhythm = extract_style(flat, mode, bpm, pitched, instr, unpitched, fast, mids)
    info = song_info(P.sub('song_info_model'), style, rhythm, fast)
    xp = pitched_style_applier(P.sub('pitched_style_applier'), style, melody, rhythm, instr)
    xu = None
    if unpitched is not None:
        xu = unpitched_style_applier(P.sub('unpitched_style_applier'), style, rhythm)
    if mids is not None:
        mids.update(style=style, melody=melody, rhythm=rhythm)
    return info, xp, xu


# ---------------------------------------------------------------- losses (style/model.py:847-997)
def _safe_div(n, d):
    """style/model.py:854-860."""
    if d.abs() < EPS:
        d = d - EPS if d < 0 else d + EPS
    return n / d


def _safe_sqrt(x):
    """style/utils/pytorch.py:68-71."""
    if x == 0:
        return torch.tensor(0., requires_grad=x.requires_grad) * x
    return torch.sqrt(x)


def qmean(ts, ws=None):
    """Quadratic mean, style/utils/pytorch.py:74-94 (weights may be live tensors)."""
    if ws is None:
        ws = [1.0 / len(ts)] * len(ts)
    return _safe_sqrt(sum(w * t * t for t, w in zip(ts, ws)))


def channels_losses(pred, target, pitched=True):
    """style/model.py:909-921 (+ :847-851, :863-896)."""
    tv, pv = target[..., 1], pred[..., 1]
    mask = (tv > 0).float()
    tp = torch.min(pv, tv).sum()
    fp = torch.relu(pv - tv).sum()
    fn = torch.relu(tv - pv).sum()
    precision = _safe_div(tp, tp + fp)
    recall = _safe_div(tp, tp + fn)
    notes = 1.0 - 2.0 * _safe_div(precision * recall, precision + recall)
    n = mask.sum()
    velocity = (((tv - pv) ** 2) * mask).sum() / n
    duration = ((((pred[..., 0] - target[..., 0].clamp(max=6)) / 6) ** 2) * mask).sum() / n
    if not pitched:
        return notes, velocity, duration
    bce = F.binary_cross_entropy(pred[..., 2:], target[..., 2:], reduction='none')
    accidentals = (bce * mask[..., None]).sum() / (n * 3)
    return notes, velocity, duration, accidentals


def merge_channel_losses(notes, velocity, duration, accidentals=None):
    """style/model.py:924-932."""
    nv = qmean([notes, velocity], [notes, 1 - notes])
    if accidentals is None:
        return qmean([duration, nv])
    return qmean([duration, accidentals, nv])


def total_loss(instr_pred, instr_target, bpm_pred, bpm_target, mode_pred, mode_target,
               pitched_pred, pitched_target, unpitched_pred=None, unpitched_target=None,
               normalize=True):
    """style/model.py:935-997 with SEMANTIC argument names: the reference's third/fourth
    positional slots are consumed as (bpm_pred, bpm_target) and the fifth/sixth as
    (mode_pred, mode_target) because of the pack/unpack swap at :900-901 vs :976-977,
    which is how train-model.py:115-122 calls it. Returns the flattened loss dict."""
    out = {}
    n, v, d, a = channels_losses(pitched_pred, pitched_target)
    if normalize:
        a = torch.tanh(a)
    p_total = merge_channel_losses(n, v, d, a)
    out.update(channels_loss_pitched_total=p_total, channels_loss_pitched_notes_loss=n,
               channels_loss_pitched_velocity_loss=v, channels_loss_pitched_duration_loss=d,
               channels_loss_pitched_accidentals_loss=a)
    if unpitched_target is not None:
        un, uv, ud = channels_losses(unpitched_pred, unpitched_target, pitched=False)
        u_total = merge_channel_losses(un, uv, ud)
        out.update(channels_loss_unpitched_total=u_total, channels_loss_unpitched_notes_loss=un,
                   channels_loss_unpitched_velocity_loss=uv, channels_loss_unpitched_duration_loss=ud)
        ch_total = qmean([p_total, u_total])
    else:
        ch_total = p_total
    out['channels_loss_total'] = ch_total

    il = F.binary_cross_entropy_with_logits(instr_pred, instr_target)        # :903
    ml = F.cross_entropy(mode_pred, mode_target.argmax(1))                   # :904
    bl = ((bpm_pred - bpm_target) / BPM_RANGE) ** 2                          # :905, shape (1,)
    if normalize:
        il, ml = torch.tanh(il), torch.tanh(ml)
    si_total = qmean([il, ml, bl])
    out.update(song_info_loss_instruments_loss=il, song_info_loss_mode_loss=ml,
               song_info_loss_bpm_loss=bl, song_info_loss_total=si_total)
    out['total'] = qmean([ch_total, si_total])
    return out


def hard_output(x):
    """style/model.py:818-832.  Like the reference (:822, `velocity *= ...` on a view) it zeroes the
    sub-threshold velocities of its INPUT in place; pinned by tests/golden/inference_small.npz."""
    vel = x[..., 1:2]
    vel *= (vel > .01).float()
    if x.shape[-1] > 2:
        acc = x[..., 2:]
        hard = ((acc == acc.max(-1, keepdim=True)[0]) & (acc > .1)).float()
        return torch.cat([x[..., :1], vel, hard], -1)
    return torch.cat([x[..., :1], vel], -1)


# ---------------------------------------------------------------- train step (train-model.py:89-90,113-126,151-154)
def iteration(flat, clip, fast=False, mids=None):
    """forward + total loss + backward (grads ACCUMULATE into .grad, no averaging)."""
    info, xp, xu = forward(flat, clip['mode'], clip['bpm'], clip['pitched'],
                           clip['instruments_features'], clip['unpitched'], fast, mids)
    losses = total_loss(info[0], clip['used_instruments'], info[2], clip['bpm_int'],
                        info[1], clip['mode'], xp, clip['pitched'], xu, clip['unpitched'])
    losses['total'].backward()
    return (info, xp, xu), {k: float(v.detach()) for k, v in losses.items()}


class Adam:
    """torch.optim.Adam(lr=.01, betas=(.9,.999), eps=1e-8) + StepLR(200,.9), restated
    (train-model.py:89-90). One `step()` = optimizer.step(); zero_grad(); scheduler.step()."""

    def __init__(self, params, lr=.01, b1=.9, b2=.999, eps=1e-8, step_size=200, gamma=.9):
        self.params = list(params)
        self.lr0, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.step_size, self.gamma = step_size, gamma
        self.t = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    def lr(self):
        return self.lr0 * self.gamma ** (self.t // self.step_size)

    @torch.no_grad()
    def step(self):
        lr = self.lr()
        self.t += 1
        c1 = 1 - self.b1 ** self.t
        c2 = 1 - self.b2 ** self.t
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(c2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-lr / c1)
            p.grad = None
