"""MI355X-native `style.model`: the reference's Python surface (style/model.py:28-997) over the HIP
hot path in libmst_amd.so.

What is kept, so that train-model.py:15-28,54-126 and style/style_transfer.py:17,67-131 run unchanged:
the nine module classes with their constructor signatures and the same `nn.Linear / nn.Conv1d /
LSTM / Distributed` children under the same attribute names (=> identical state_dict keys,
parameter order, seed-108 initialisation and pickle class paths), `StyleTransferModel` with
`extract_style / predict_song_info / apply_style / forward`, `get_total_loss` (including the
positional bpm-before-mode contract of train-model.py:115-122), `hard_output`, `device`.

What is different: no module executes torch ops.  The children are parameter containers; the
four model methods and the loss are `torch.autograd.Function`s that borrow `data_ptr()`s and
enqueue hand-written HIP kernels through the C ABI (include/mst_amd.h).  There is no CPU
fallback: without the library or without a GPU tensor the calls raise.
"""
import collections
import math

import numpy as np
import torch
from torch import nn

from style import _native
from style.utils.pytorch import Distributed, LSTM

epsilon = 1e-7
n_beat_fractions = 10
n_pitched_features = 5
n_unpitched_features = 2
n_octaves = 8
n_scale_degrees = 7
n_pitched_notes = n_octaves * n_scale_degrees
n_unpitched_notes = 47
n_modes = 2
min_bpm = 50
max_bpm = 200
bpm_range = max_bpm - min_bpm
mean_type = 'quadratic'
device = 'cuda' if torch.cuda.is_available() else 'cpu'

_DEFAULT_WIDTHS = dict(beat=64, bar=128, nrf=8, style=256, melody=8, rhythm=32, instr=51, n_instruments=41)
# Distributed(depth) of the wrapped children (style/model.py:53,68,119,467)
_DEPTH = {('pitched_channels_encoder', 'beats_conv'): 3, ('pitched_channels_encoder', 'beats_lstm'): 2,
          ('unpitched_channels_encoder', 'beats_lstm'): 2, ('song_info_model', 'beats_lstm'): 1}


def get_mean_size(*values, factor=1):
    return math.ceil(np.mean(values) * factor)


def _dims(C=1, R=1, T=1, unpitched=True, **widths):
    w = dict(_DEFAULT_WIDTHS)
    w.update(widths)
    return _native.Dims(C=C, R=R, T=T, beat=w['beat'], bar=w['bar'], nrf=w['nrf'], style=w['style'], melody=w['melody'],
                        rhythm=w['rhythm'], instr=w['instr'], n_instruments=w['n_instruments'], has_unpitched=int(unpitched))


class _Container(nn.Module):
    """A sub-module of the reference as a parameter container.  Children are created from the C
    ABI's parameter table (mst_param_info), in its order, so Python and HIP agree by construction."""
    _prefix = None

    def _build(self, **widths):
        self._widths = widths
        table = _native.get().param_table(_dims(**widths))
        groups = collections.OrderedDict()
        for name, _, shape in table:
            if not name.startswith(self._prefix + '.'):
                continue
            rest = name[len(self._prefix) + 1:]
            child, leaf = rest.split('.', 1)
            groups.setdefault(child, {})[leaf] = shape
        for child, leaves in groups.items():
            wrapped = any(k.startswith('module.') for k in leaves)
            leaves = {k[len('module.'):] if wrapped else k: v for k, v in leaves.items()}
            if 'weight_ih_l0' in leaves:
                four_h, n_in = leaves['weight_ih_l0']
                mod = LSTM(input_size=n_in, hidden_size=four_h // 4, num_layers=1, batch_first=True,
                           bidirectional='weight_ih_l0_reverse' in leaves)
            elif len(leaves['weight']) == 3:
                oc, ic, k = leaves['weight']
                mod = nn.Conv1d(in_channels=ic, out_channels=oc, kernel_size=k, stride=n_scale_degrees, padding=4)
            else:
                n_out, n_in = leaves['weight']
                mod = nn.Linear(in_features=n_in, out_features=n_out)
            if wrapped:
                mod = Distributed(mod, depth=_DEPTH[(self._prefix, child)])
            setattr(self, child, mod)

    def _owner(self):
        ref = self.__dict__.get('_parent')
        model = ref() if ref is not None else None
        if model is None:
            raise NotImplementedError(
                f'{type(self).__name__} is a parameter container on the MI355X path: its arithmetic is fused into the HIP kernels '
                'behind StyleTransferModel.extract_style / predict_song_info / apply_style / forward, and it can only be called '
                'on its own once it is part of a StyleTransferModel (the kernels need the whole flat parameter buffer).')
        return model

    def forward(self, *args, **kwargs):
        self._owner()
        raise NotImplementedError(
            f'{type(self).__name__}.forward takes intermediate tensors (beats / bars) that the fused plan does not accept as inputs; '
            'call StyleTransferModel.extract_style instead.  Stand-alone forwards exist for the channel encoders, SongInfoModel and '
            'the two style appliers (INTEGRATION.md section 2).')

    def __getstate__(self):          # the back-reference to the owning model is rebuilt by StyleTransferModel, never pickled
        d = dict(self.__dict__)
        d.pop('_parent', None)
        return d


def _named_slots(model, plan, names, shapes):
    return tuple(plan.view(n, s).clone() for n, s in zip(names, shapes))


class PitchedChannelsEncoder(_Container):
    _prefix = 'pitched_channels_encoder'

    def __init__(self, beat_size, bar_size, instrument_size):
        super().__init__()
        assert bar_size % 2 == 0
        self._build(beat=beat_size, bar=bar_size, instr=instrument_size)

    def forward(self, x, instruments_features):
        """style/model.py:77-99 on its own (forward only, no autograd): (beats (1,C,R,T,beat), bars (1,R,bar)).  Runs the
        extract stage of the owning model's plan and returns its named slots."""
        model = self._owner()
        with torch.no_grad():
            dev = model._anchor().device
            x = _f32c(x, dev)
            _, C, R, T = x.shape[:4]
            plan = model._plan(C, R, T, False, dev)
            plan.set_inputs(mode=torch.tensor([1., 0.]), bpm=torch.tensor([120.]), instr=_f32c(instruments_features, dev))
            plan.forward(_native.STAGE_EXTRACT, model._flat, x, None)
            return _named_slots(model, plan, ('pitched_beats', 'pitched_bars'), ((1, C, R, T, -1), (1, R, -1)))


class UnpitchedChannelsEncoder(_Container):
    _prefix = 'unpitched_channels_encoder'

    def __init__(self, beat_size, bar_size):
        super().__init__()
        assert bar_size % 2 == 0
        self._build(beat=beat_size, bar=bar_size)

    def forward(self, x):
        """style/model.py:128-141 on its own (forward only): (beats (1,1,R,T,beat), bars (1,R,bar)).  The plan needs a pitched
        tensor beside it; an all-zero single channel is supplied (the unpitched encoder does not read it)."""
        model = self._owner()
        with torch.no_grad():
            dev = model._anchor().device
            x = _f32c(x, dev)
            _, _, R, T = x.shape[:4]
            plan = model._plan(1, R, T, True, dev)
            pitched = torch.zeros(1, 1, R, T, n_beat_fractions, n_pitched_notes, n_pitched_features, device=dev)
            plan.set_inputs(mode=torch.tensor([1., 0.]), bpm=torch.tensor([120.]), instr=torch.zeros(1, model._widths.get('instr', 51)))
            plan.forward(_native.STAGE_EXTRACT, model._flat, pitched, x)
            return _named_slots(model, plan, ('unpitched_beats', 'unpitched_bars'), ((1, 1, R, T, -1), (1, R, -1)))


class StyleEncoder(_Container):
    _prefix = 'style_encoder'

    def __init__(self, style_size, bar_size, instrument_size):
        super().__init__()
        self._build(style=style_size, bar=bar_size, instr=instrument_size)


class MelodyEncoder(_Container):
    _prefix = 'melody_encoder'

    def __init__(self, melody_size, beat_size, bar_size, instrument_size):
        super().__init__()
        self._build(melody=melody_size, beat=beat_size, bar=bar_size, instr=instrument_size)


class PitchedRhythmEncoder(_Container):
    _prefix = 'pitched_rhythm_encoder'

    def __init__(self, rhythm_size, beat_size, bar_size, instrument_size):
        super().__init__()
        self._build(rhythm=rhythm_size, beat=beat_size, bar=bar_size, instr=instrument_size)


class UnpitchedRhythmEncoder(_Container):
    _prefix = 'unpitched_rhythm_encoder'

    def __init__(self, rhythm_size, beat_size, bar_size):
        super().__init__()
        self._build(rhythm=rhythm_size, beat=beat_size, bar=bar_size)


class SongInfoModel(_Container):
    _prefix = 'song_info_model'

    def __init__(self, n_rhythm_features, style_size, rhythm_size, n_instruments):
        super().__init__()
        self._build(nrf=n_rhythm_features, style=style_size, rhythm=rhythm_size, n_instruments=n_instruments)

    def forward(self, style, rhythm):
        """style/model.py:557-562 = StyleTransferModel.predict_song_info (differentiable)."""
        return self._owner().predict_song_info(style, rhythm)


class PitchedStyleApplier(_Container):
    _prefix = 'pitched_style_applier'

    def __init__(self, style_size, melody_size, rhythm_size, instrument_size):
        super().__init__()
        self._build(style=style_size, melody=melody_size, rhythm=rhythm_size, instr=instrument_size)

    def forward(self, style, melody, rhythm, instruments):
        """style/model.py:624-675 = the pitched half of StyleTransferModel.apply_style (differentiable)."""
        return self._owner().apply_style(style, melody, rhythm, instruments, unpitched=False)[0]


class UnpitchedStyleApplier(_Container):
    _prefix = 'unpitched_style_applier'

    def __init__(self, style_size, rhythm_size):
        super().__init__()
        self._build(style=style_size, rhythm=rhythm_size)

    def forward(self, style, rhythm):
        """style/model.py:703-724 (forward only): the unpitched half of apply_style; a zero melody and one blank instrument row
        stand in for the pitched applier's inputs, which this module does not read."""
        model = self._owner()
        with torch.no_grad():
            R, T = rhythm.shape[1:3]
            w = dict(_DEFAULT_WIDTHS); w.update(model._widths)
            melody = torch.zeros(1, R, T, n_beat_fractions, n_pitched_notes, w['melody'], device=rhythm.device)
            instr = torch.zeros(1, 1, w['instr'], device=rhythm.device)
            return model.apply_style(style, melody, rhythm, instr, unpitched=True)[1]


def _f32c(t, dev):
    return torch.as_tensor(t, dtype=torch.float32, device=dev).contiguous()


class StyleTransferModel(nn.Module):
    def __init__(self, pitched_channels_encoder, unpitched_channels_encoder, style_encoder, melody_encoder,
                 pitched_rhythm_encoder, unpitched_rhythm_encoder, song_info_model, pitched_style_applier,
                 unpitched_style_applier):
        super().__init__()
        self.pitched_channels_encoder = pitched_channels_encoder
        self.unpitched_channels_encoder = unpitched_channels_encoder
        self.style_encoder = style_encoder
        self.melody_encoder = melody_encoder
        self.pitched_rhythm_encoder = pitched_rhythm_encoder
        self.unpitched_rhythm_encoder = unpitched_rhythm_encoder
        self.song_info_model = song_info_model
        self.pitched_style_applier = pitched_style_applier
        self.unpitched_style_applier = unpitched_style_applier
        widths = {}
        for _, child in self.named_children():
            for k, v in child._widths.items():
                if widths.setdefault(k, v) != v:
                    raise ValueError(f'sub-modules disagree on {k}_size: {widths[k]} vs {v}')
        self._widths = widths
        self._link_children()
        # fail here, not at the first forward: the note-level HIP kernels exist for melody_size 8 and 4 only
        if _native.get().lib.mst_widths_supported(_dims(**widths)) != 0:
            raise _native.MstError(f'layer widths {widths} are outside the instantiated HIP kernels '
                                   '(melody_size must be 8 or 4; LSTM hidden sizes <= 256)')
        self._flat = self._gflat = None
        self._offsets = self._ends = self._slots = None
        self.graph_repeated_shapes = True       # train_iteration(): replay a hipGraph when a clip shape comes back

    def _link_children(self):
        import weakref
        for _, child in self.named_children():       # lets a sub-module be called on its own (see _Container.forward)
            child.__dict__['_parent'] = weakref.ref(self)

    def __setstate__(self, state):                   # whole-module snapshots (train-model.py:156-160): re-link after unpickling
        super().__setstate__(state)
        self._link_children()

    # ---- flat parameter / gradient buffers -------------------------------------------------
    def _sync_flat(self):
        """Point every Parameter's storage into ONE flat fp32 buffer laid out as the C ABI expects
        (model.parameters() order).  Re-done if the parameters moved (.to(), load_state_dict of new tensors)."""
        # fast path (every forward of a training loop): every Parameter OBJECT registered in the module tree is still the one
        # that was aliased into the flat buffer (identity against the leaf modules' own `_parameters` dicts: catches
        # load_state_dict(assign=True) and `module.weight = nn.Parameter(...)` anywhere in the tree, which install new objects
        # while the old ones keep aliasing the flat buffer), and the first and the last of them still point where they should
        # (catches .to(), which re-homes the data of all of them)
        ends = getattr(self, '_ends', None)
        slots = getattr(self, '_slots', None)
        if self._flat is not None and ends is not None and slots is not None:
            base = self._flat.data_ptr()
            if (all(reg.get(key) is p for reg, key, p in slots) and ends[0][0].data_ptr() == base + 4 * ends[0][1]
                    and ends[1][0].data_ptr() == base + 4 * ends[1][1]):
                return
        named = list(self.named_parameters())
        dev = named[0][1].device
        if dev.type != 'cuda':
            raise _native.MstError('the MI355X path needs the model on a GPU (model.to("cuda")); there is no CPU fallback')
        if self._flat is not None and self._flat.device == dev and all(
                p.data_ptr() == self._flat.data_ptr() + 4 * off for (_, p), off in zip(named, self._offsets)):
            self._ends = ((named[0][1], self._offsets[0]), (named[-1][1], self._offsets[-1]))
            self._slots = self._param_slots()
            return
        table = _native.get().param_table(_dims(**self._widths))
        if [n for n, _ in named] != [n for n, _, _ in table]:
            raise _native.MstError('parameter names/order differ from the C ABI layout')
        flat = torch.empty(_native.get().param_floats(_dims(**self._widths)), dtype=torch.float32, device=dev)
        for (name, p), (_, off, shape) in zip(named, table):
            if tuple(p.shape) != shape:
                raise _native.MstError(f'{name}: shape {tuple(p.shape)} != {shape}')
            flat[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[off:off + p.numel()].view(shape)
        self._flat, self._gflat = flat, torch.zeros_like(flat)
        self._offsets = [off for _, off, _ in table]
        self._ends = ((named[0][1], self._offsets[0]), (named[-1][1], self._offsets[-1]))
        self._slots = self._param_slots()

    def _param_slots(self):
        """(leaf module's `_parameters` dict, key, Parameter) for every parameter, in model.parameters() order."""
        return [(mod._parameters, key, p) for mod in self.modules() for key, p in mod._parameters.items() if p is not None]

    def _grad_target(self):
        """Where backward accumulates: the flat gradient buffer, aliased by every p.grad."""
        (p0, o0), (p1, o1) = self._ends            # fast path: first and last p.grad alias the flat gradient buffer
        gb = self._gflat.data_ptr()
        if p0.grad is not None and p1.grad is not None and p0.grad.data_ptr() == gb + 4 * o0 and p1.grad.data_ptr() == gb + 4 * o1:
            return self._gflat
        params = list(self.parameters())
        mine = [p.grad is not None and p.grad.data_ptr() == self._gflat.data_ptr() + 4 * off
                for p, off in zip(params, self._offsets)]
        if not any(p.grad is not None for p in params):
            self._gflat.zero_()             # zero_grad(set_to_none=True) dropped the aliases
        elif not all(mine):
            raise _native.MstError('p.grad was replaced by foreign tensors; use zero_grad() or keep the aliased grads')
        return self._gflat

    def _publish_grads(self):
        if self._ends[0][0].grad is not None and self._ends[1][0].grad is not None:
            return
        for p, off in zip(self.parameters(), self._offsets):
            if p.grad is None:
                p.grad = self._gflat[off:off + p.numel()].view(p.shape)

    def _plan(self, C, R, T, unpitched, dev):
        return _native.get().plan(_dims(C=C, R=R, T=T, unpitched=unpitched, **self._widths), dev)

    def _anchor(self):
        self._sync_flat()
        return next(self.parameters())

    # ---- reference surface (style/model.py:751-793) ----------------------------------------
    # Grad mode is read HERE: inside autograd.Function.forward it is always off, so the Functions take it as a plain bool.
    def extract_style(self, mode, bpm, pitched_channels, instruments_features, unpitched_channels=None):
        return _Extract.apply(self, self._anchor(), torch.is_grad_enabled(), mode, bpm, pitched_channels, instruments_features,
                              unpitched_channels)

    def predict_song_info(self, style, rhythm):
        return _Predict.apply(self, self._anchor(), torch.is_grad_enabled(), style, rhythm)

    def apply_style(self, style, melody, rhythm, instruments_features, unpitched=False):
        x_pitched, x_unpitched = _Apply.apply(self, self._anchor(), torch.is_grad_enabled(), style, melody, rhythm,
                                              instruments_features, bool(unpitched))
        return x_pitched, (x_unpitched if unpitched else None)

    def train_iteration(self, mode, bpm, pitched_channels, instruments_features, unpitched_channels, used_instruments, bpm_target):
        """Opt-in fast form of one train-model.py loop body (train-model.py:113-126): forward, get_total_loss(normalize=True)
        with the inputs as targets, and loss.backward() in ONE C-ABI call (mst_train_iteration) — no autograd graph, no
        per-stage Python.  Gradients accumulate (sum) for optimizer.step() as usual.  Returns the 15 loss leaves as one device
        tensor (key order style._native.LOSS_KEYS; a row of a ring, valid for the next 256 calls of its lane).

        With `concurrent_accumulation` (FusedAdam switches it on) consecutive calls alternate between two LANES — two side
        streams, two workspaces per plan, two gradient buffers — so the iter_size = 2 accumulation iterations between two
        optimizer steps (train-model.py:95,151-153; they are independent: same parameters, gradients summed) overlap on the
        device; FusedAdam.step() joins the lanes and applies Adam to the sum of the two buffers (mst_adam_step2: bitwise what
        in-place accumulation gives).  p.grad shows lane 0's share until then."""
        anchor = self._anchor()
        dev = anchor.device
        pitched = _f32c(pitched_channels, dev)
        unpitched = None if unpitched_channels is None else _f32c(unpitched_channels, dev)
        _, C, R, T = pitched.shape[:4]
        plan = self._plan(C, R, T, unpitched is not None, dev)
        lanes = self._lanes(dev)
        li = 0
        if getattr(self, 'concurrent_accumulation', False):
            li = self._lane_next
            self._lane_next ^= 1
        lane = lanes[li]
        gflat = self._grad_target() if li == 0 else lane['grad']
        st = plan.__dict__.setdefault('_lane_state', {}).get(li)
        if st is None:
            st = plan._lane_state[li] = dict(ws=plan.ws if li == 0 else plan.new_ws(), graph=None, key=None, uses=0,
                                             pitched=torch.empty_like(pitched), unpitched=None if unpitched is None else torch.empty_like(unpitched),
                                             losses=torch.empty(_native.N_LOSSES, dtype=torch.float32, device=dev))
        ws = st['ws']
        small = [_f32c(mode, dev).reshape(-1), _f32c(bpm, dev).reshape(-1), _f32c(instruments_features, dev).reshape(-1),
                 _f32c(used_instruments, dev).reshape(-1),
                 torch.as_tensor(float(bpm_target), dtype=torch.float32).reshape(1).to(dev, non_blocking=True)]
        slots = [plan.view(n, ws=ws) for n in ('mode', 'bpm', 'instr', 'used_instruments', 'bpm_target')]
        cur = torch.cuda.current_stream(dev)
        side = lane['stream'] if li or getattr(self, 'concurrent_accumulation', False) else cur
        if side is not cur:
            side.wait_stream(cur)                      # the caller's inputs (and the last optimizer step) are ready
            for t in small + [pitched] + ([unpitched] if unpitched is not None else []):
                t.record_stream(side)
        with torch.cuda.stream(side):
            # the note tensors are copied into the lane's static buffers unless the caller already wrote them there
            # (static_inputs()): a captured graph reads fixed addresses
            srcs, dsts = list(small), list(slots)
            if pitched.data_ptr() != st['pitched'].data_ptr():
                srcs.append(pitched.reshape(-1)); dsts.append(st['pitched'].reshape(-1))
            if unpitched is not None and unpitched.data_ptr() != st['unpitched'].data_ptr():
                srcs.append(unpitched.reshape(-1)); dsts.append(st['unpitched'].reshape(-1))
            torch._foreach_copy_(dsts, srcs)           # one launch for all of them
            # A shape seen before replays a hipGraph of the whole loop body (the launches of mst_train_iteration captured once
            # per plan and lane over static input buffers); songs of a new shape run eagerly.  Keyed by the buffers it baked in.
            key = (self._flat.data_ptr(), gflat.data_ptr())
            g = st['graph'] if st['key'] == key else None
            if g is None and st['uses'] >= 1 and dev.type == 'cuda' and getattr(self, 'graph_repeated_shapes', True):
                # (the plan has run eagerly before, so its code objects are loaded: capturing executes nothing and needs no warm-up)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side if side is not cur else None):
                    plan.train_iteration(self._flat, gflat, st['pitched'], st['unpitched'], st['losses'], ws=ws)
                st['graph'], st['key'] = g, key
            st['uses'] += 1
            if g is not None:
                plan._touch(ws)                         # a replay launches on this workspace
                g.replay()
            else:
                plan.train_iteration(self._flat, gflat, st['pitched'], st['unpitched'], st['losses'], ws=ws)
            ring = lane['ring']
            row = ring[lane['at'] % ring.shape[0]]
            lane['at'] += 1
            row.copy_(st['losses'])
        lane['dirty'] = True
        if li == 0:
            self._publish_grads()
        return row

    def static_inputs(self, C, R, T, unpitched=True, lane=0):
        """The note-tensor buffers a captured train_iteration of this shape reads (created on first use by train_iteration): a
        loader that writes its H2D copies straight into them saves the per-iteration device copy."""
        plan = self._plan(C, R, T, unpitched, self._anchor().device)
        st = plan.__dict__.get('_lane_state', {}).get(lane)
        return (st['pitched'], st['unpitched']) if st else None

    def _lanes(self, dev):
        lanes = self.__dict__.get('_lane_list')
        if lanes is None or lanes[0]['ring'].device != dev:
            mk = lambda: dict(stream=torch.cuda.Stream(dev) if dev.type == 'cuda' else None, grad=None, dirty=False, at=0,
                              ring=torch.zeros(256, _native.N_LOSSES, dtype=torch.float32, device=dev))
            lanes = self.__dict__['_lane_list'] = [mk(), mk()]
            lanes[1]['grad'] = torch.zeros_like(self._flat)
            self.__dict__['_lane_next'] = 0
        if lanes[1]['grad'].data_ptr() == 0 or lanes[1]['grad'].numel() != self._flat.numel() or lanes[1]['grad'].device != self._flat.device:
            lanes[1]['grad'] = torch.zeros_like(self._flat)
        return lanes

    def join_lanes(self):
        """Make the current stream wait for everything train_iteration has enqueued on its lanes; returns lane 1's gradient
        buffer if it holds a share of the accumulated gradient (else None).  FusedAdam.step and LossLog.flush call it."""
        lanes = self.__dict__.get('_lane_list')
        if not lanes:
            return None
        cur = torch.cuda.current_stream(lanes[0]['ring'].device)
        for lane in lanes:
            if lane['dirty'] and lane['stream'] is not None:
                cur.wait_stream(lane['stream'])
        second = lanes[1]['grad'] if lanes[1]['dirty'] else None
        for lane in lanes:
            lane['dirty'] = False
        self.__dict__['_lane_next'] = 0
        return second

    def __getstate__(self):                          # lane streams / rings / graphs are run-time state, never pickled
        d = dict(self.__dict__)
        for k in ('_lane_list', '_lane_next'):
            d.pop(k, None)
        return d

    def check_device_status(self):
        """Raise MstError if a kernel reported a device-side failure (include/mst_amd.h MST_DEV_*: the multi-workgroup LSTM's
        exchange timing out) on any workspace used since the last check.  Synchronises; call it where the loop reads its
        losses back anyway (style.train.LossLog.flush does)."""
        self.join_lanes_keep()
        for plan in list(_native.get()._plans.values()):
            plan.check_touched()

    def join_lanes_keep(self):
        """Wait for the lanes without consuming their gradient bookkeeping (a health check is not an optimizer step)."""
        lanes = self.__dict__.get('_lane_list')
        if lanes:
            cur = torch.cuda.current_stream(lanes[0]['ring'].device)
            for lane in lanes:
                if lane['dirty'] and lane['stream'] is not None:
                    cur.wait_stream(lane['stream'])

    def forward(self, mode, bpm, pitched_channels, instruments_features, unpitched_channels=None):
        ip, mp, bp, xp, xu = _Forward.apply(self, self._anchor(), torch.is_grad_enabled(), mode, bpm, pitched_channels,
                                            instruments_features, unpitched_channels)
        return (ip, mp, bp), xp, (xu if unpitched_channels is not None else None)


def _shapes(model, C, R, T):
    w = dict(_DEFAULT_WIDTHS)
    w.update(model._widths)
    return dict(style=(1, w['style']), melody=(1, R, T, 10, 56, w['melody']), rhythm=(1, R, T, 10, w['rhythm']),
                instruments_pred=(1, w['n_instruments']), mode_pred=(1, 2), bpm_pred=(1,),
                pitched_pred=(1, C, R, T, 10, 56, 5), unpitched_pred=(1, 1, R, T, 10, 47, 2))


def _seed(plan, ws, name, g):
    slot = plan.grad(name, ws=ws)
    if g is None:
        slot.zero_()
    else:
        slot.copy_(g.reshape(-1))


class _StageFn(torch.autograd.Function):
    """Common plumbing: one workspace per forward whose backward is pending (two grad-enabled forwards of the same
    shape may both be waiting for their backward, e.g. the summed loss of two clips); it comes from the plan's pool
    and goes back when that backward has run.  Without grad the plan's own scratch workspace is reused."""

    @staticmethod
    def _ws(plan, grad):
        return plan.acquire_ws() if grad else plan.ws

    @staticmethod
    def _take(ctx):
        """The saved state of a forward, exactly once: its workspace goes back to the pool after this backward."""
        stuff, ctx.stuff = ctx.stuff, None
        if stuff is None:
            raise _native.MstError('this stage was already back-propagated (its workspace has been recycled); '
                                   'run the forward again instead of retain_graph=True')
        return stuff


class _Extract(_StageFn):
    @staticmethod
    def forward(ctx, model, anchor, grad, mode, bpm, pitched, instr, unpitched):
        dev = anchor.device
        pitched = _f32c(pitched, dev)
        unpitched = None if unpitched is None else _f32c(unpitched, dev)
        _, C, R, T = pitched.shape[:4]
        plan = model._plan(C, R, T, unpitched is not None, dev)
        ws = _StageFn._ws(plan, grad)
        plan.set_inputs(mode=_f32c(mode, dev), bpm=_f32c(bpm, dev), instr=_f32c(instr, dev), ws=ws)
        plan.forward(_native.STAGE_EXTRACT, model._flat, pitched, unpitched, ws=ws)
        shp = _shapes(model, C, R, T)
        ctx.stuff = (model, plan, ws, pitched, unpitched)
        return tuple(plan.view(k, shp[k], ws=ws).clone() for k in ('style', 'melody', 'rhythm'))

    @staticmethod
    def backward(ctx, g_style, g_melody, g_rhythm):
        model, plan, ws, pitched, unpitched = _StageFn._take(ctx)
        plan.zero_grads(_native.STAGE_EXTRACT, ws=ws)
        for k, g in (('style', g_style), ('melody', g_melody), ('rhythm', g_rhythm)):
            _seed(plan, ws, k, g)
        plan.backward(_native.STAGE_EXTRACT, model._flat, model._grad_target(), pitched, unpitched, ws=ws)
        model._publish_grads()
        plan.release_ws(ws)
        return (None,) * 8


class _Predict(_StageFn):
    @staticmethod
    def forward(ctx, model, anchor, grad, style, rhythm):
        dev = anchor.device
        R, T = rhythm.shape[1:3]
        plan = model._plan(1, R, T, False, dev)
        ws = _StageFn._ws(plan, grad)
        plan.view('style', ws=ws).copy_(_f32c(style, dev).reshape(-1))
        plan.view('rhythm', ws=ws).copy_(_f32c(rhythm, dev).reshape(-1))
        plan.forward(_native.STAGE_INFO, model._flat, None, None, ws=ws)
        shp = _shapes(model, 1, R, T)
        ctx.stuff = (model, plan, ws)
        return tuple(plan.view(k, shp[k], ws=ws).clone() for k in ('instruments_pred', 'mode_pred', 'bpm_pred'))

    @staticmethod
    def backward(ctx, g_instr, g_mode, g_bpm):
        model, plan, ws = _StageFn._take(ctx)
        plan.zero_grads(_native.STAGE_INFO, ws=ws)
        for k in ('style', 'rhythm'):
            plan.grad(k, ws=ws).zero_()
        for k, g in (('instruments_pred', g_instr), ('mode_pred', g_mode), ('bpm_pred', g_bpm)):
            _seed(plan, ws, k, g)
        plan.backward(_native.STAGE_INFO, model._flat, model._grad_target(), None, None, ws=ws)
        model._publish_grads()
        shp = _shapes(model, 1, *[int(v) for v in (plan.dims.R, plan.dims.T)])
        out = (None, None, None, plan.grad('style', shp['style'], ws=ws).clone(), plan.grad('rhythm', shp['rhythm'], ws=ws).clone())
        plan.release_ws(ws)
        return out


class _Apply(_StageFn):
    @staticmethod
    def forward(ctx, model, anchor, grad, style, melody, rhythm, instr, unpitched):
        dev = anchor.device
        instr = _f32c(instr, dev)
        C = instr.shape[1]
        R, T = rhythm.shape[1:3]
        plan = model._plan(C, R, T, unpitched, dev)
        ws = _StageFn._ws(plan, grad)
        plan.set_inputs(instr=instr, ws=ws)
        for k, t in (('style', style), ('melody', melody), ('rhythm', rhythm)):
            plan.view(k, ws=ws).copy_(_f32c(t, dev).reshape(-1))
        plan.forward(_native.STAGE_APPLY, model._flat, None, None, ws=ws)
        shp = _shapes(model, C, R, T)
        ctx.stuff = (model, plan, ws, unpitched)
        xp = plan.view('pitched_pred', shp['pitched_pred'], ws=ws).clone()
        xu = plan.view('unpitched_pred', shp['unpitched_pred'], ws=ws).clone() if unpitched else xp.new_zeros(1)
        return xp, xu

    @staticmethod
    def backward(ctx, g_xp, g_xu):
        model, plan, ws, unpitched = _StageFn._take(ctx)
        plan.zero_grads(_native.STAGE_APPLY, ws=ws)
        for k in ('style', 'melody', 'rhythm'):
            plan.grad(k, ws=ws).zero_()
        _seed(plan, ws, 'pitched_pred', g_xp)
        if unpitched:
            _seed(plan, ws, 'unpitched_pred', g_xu)
        plan.backward(_native.STAGE_APPLY, model._flat, model._grad_target(), None, None, ws=ws)
        model._publish_grads()
        shp = _shapes(model, plan.dims.C, plan.dims.R, plan.dims.T)
        out = (None, None, None) + tuple(plan.grad(k, shp[k], ws=ws).clone() for k in ('style', 'melody', 'rhythm')) + (None, None)
        plan.release_ws(ws)
        return out


class _Forward(_StageFn):
    """extract_style + predict_song_info + apply_style in one workspace (the training path)."""

    @staticmethod
    def forward(ctx, model, anchor, grad, mode, bpm, pitched, instr, unpitched):
        dev = anchor.device
        pitched = _f32c(pitched, dev)
        unpitched = None if unpitched is None else _f32c(unpitched, dev)
        _, C, R, T = pitched.shape[:4]
        U = unpitched is not None
        plan = model._plan(C, R, T, U, dev)
        ws = _StageFn._ws(plan, grad)
        plan.set_inputs(mode=_f32c(mode, dev), bpm=_f32c(bpm, dev), instr=_f32c(instr, dev), ws=ws)
        plan.forward(_native.STAGE_ALL, model._flat, pitched, unpitched, ws=ws)
        shp = _shapes(model, C, R, T)
        ctx.stuff = (model, plan, ws, pitched, unpitched)
        keys = ['instruments_pred', 'mode_pred', 'bpm_pred', 'pitched_pred']
        outs = [plan.view(k, shp[k], ws=ws).clone() for k in keys]
        outs.append(plan.view('unpitched_pred', shp['unpitched_pred'], ws=ws).clone() if U else outs[0].new_zeros(1))
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_instr, g_mode, g_bpm, g_xp, g_xu):
        model, plan, ws, pitched, unpitched = _StageFn._take(ctx)
        plan.zero_grads(_native.STAGE_ALL, ws=ws)
        for k, g in (('instruments_pred', g_instr), ('mode_pred', g_mode), ('bpm_pred', g_bpm), ('pitched_pred', g_xp)):
            _seed(plan, ws, k, g)
        if unpitched is not None:
            _seed(plan, ws, 'unpitched_pred', g_xu)
        plan.backward(_native.STAGE_ALL, model._flat, model._grad_target(), pitched, unpitched, ws=ws)
        model._publish_grads()
        plan.release_ws(ws)
        return (None,) * 8


# ---- losses (style/model.py:847-997) --------------------------------------------------------
class _Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pp, pt, up, ut, il, it, ml, mt, bp, bt, normalize):
        lib = _native.get().lib
        dev = pp.device
        if dev.type != 'cuda':
            raise _native.MstError('get_total_loss needs GPU tensors; there is no CPU fallback')
        pp, pt, il, it, ml, mt, bp, bt = (_f32c(t, dev) for t in (pp, pt, il, it, ml, mt, bp, bt))
        has_u = up is not None and ut is not None
        up, ut = (_f32c(up, dev), _f32c(ut, dev)) if has_u else (None, None)
        losses = torch.empty(_native.N_LOSSES, dtype=torch.float32, device=dev)
        saved = torch.empty(_native.LOSS_SAVED, dtype=torch.float32, device=dev)
        scratch = torch.empty(lib.mst_loss_scratch_floats(), dtype=torch.float32, device=dev)
        n_p = pp.numel() // 5
        n_u = up.numel() // 2 if has_u else 0
        P = _native.ptr
        _native.check(lib.mst_total_loss_fwd(P(pp), P(pt), n_p, P(up), P(ut), n_u, P(il), P(it), il.numel(), P(ml), P(mt),
                                             P(bp), P(bt), int(normalize), P(losses), P(saved), P(scratch),
                                             _native.current_stream(dev)), 'mst_total_loss_fwd')
        ctx.stuff = (pp, pt, up, ut, il, it, ml, mt, bp, bt, saved, n_p, n_u)
        return losses

    @staticmethod
    def backward(ctx, gl):
        pp, pt, up, ut, il, it, ml, mt, bp, bt, saved, n_p, n_u = ctx.stuff
        lib = _native.get().lib
        dev = pp.device
        gl = torch.nan_to_num(_f32c(gl, dev))
        g_pp, g_il, g_ml, g_bp = (torch.empty_like(t) for t in (pp, il, ml, bp))
        g_up = torch.empty_like(up) if up is not None else None
        P = _native.ptr
        _native.check(lib.mst_total_loss_bwd(P(pp), P(pt), n_p, P(up), P(ut), n_u, P(il), P(it), il.numel(), P(ml), P(mt),
                                             P(bp), P(bt), P(saved), P(gl), P(g_pp), P(g_up), P(g_il), P(g_ml), P(g_bp),
                                             _native.current_stream(dev)), 'mst_total_loss_bwd')
        return g_pp, None, g_up, None, g_il, None, g_ml, None, g_bp, None, None


class LossDict(dict):
    """The reference's nested loss dict; `.packed` is the 15-leaf device tensor behind it (key order
    `_native.LOSS_KEYS`, NaN for absent unpitched leaves) so a training loop can log without a sync per leaf."""

    def __init__(self, packed, **items):
        super().__init__(**items)
        self.packed = packed


def get_total_loss(instruments_pred, instruments_target, mode_pred, mode_target, bpm_pred, bpm_target,
                   pitched_pred, pitched_target, unpitched_pred=None, unpitched_target=None, normalize=False):
    """Same nested dict as the reference (style/model.py:944-996).  Positional contract preserved: the
    third/fourth slots are consumed as (bpm prediction, bpm target) and the fifth/sixth as (mode logits,
    one-hot mode) — the reference's pack/unpack swap at :900-901 vs :976-977, which is how
    train-model.py:115-122 calls it (bpm before mode)."""
    dev = pitched_pred.device
    bpm_p, bpm_t, mode_p, mode_t = mode_pred, mode_target, bpm_pred, bpm_target
    bpm_t = torch.as_tensor(bpm_t, dtype=torch.float32, device=dev).reshape(-1)[:1]
    L = _Loss.apply(pitched_pred, pitched_target, unpitched_pred if unpitched_target is not None else None,
                    unpitched_target, instruments_pred, instruments_target, mode_p, mode_t, bpm_p.reshape(-1), bpm_t,
                    bool(normalize))
    k = {name: i for i, name in enumerate(_native.LOSS_KEYS)}
    pitched = dict(total=L[k['channels_loss_pitched_total']], notes_loss=L[k['channels_loss_pitched_notes_loss']],
                   velocity_loss=L[k['channels_loss_pitched_velocity_loss']],
                   duration_loss=L[k['channels_loss_pitched_duration_loss']],
                   accidentals_loss=L[k['channels_loss_pitched_accidentals_loss']])
    unpitched = None
    if unpitched_target is not None:
        unpitched = dict(total=L[k['channels_loss_unpitched_total']], notes_loss=L[k['channels_loss_unpitched_notes_loss']],
                         velocity_loss=L[k['channels_loss_unpitched_velocity_loss']],
                         duration_loss=L[k['channels_loss_unpitched_duration_loss']])
    one = lambda name: L[k[name]:k[name] + 1]      # shape (1,), like the reference's bpm-derived leaves
    return LossDict(L.detach(),
        total=one('total'),
        channels_loss=dict(total=L[k['channels_loss_total']], pitched=pitched, unpitched=unpitched),
        song_info_loss=dict(total=one('song_info_loss_total'), instruments_loss=L[k['song_info_loss_instruments_loss']],
                            mode_loss=L[k['song_info_loss_mode_loss']], bpm_loss=one('song_info_loss_bpm_loss')),
    )


def hard_output(x):
    """style/model.py:818-832 — like the reference it also zeroes sub-threshold velocities of `x` in place."""
    if x.device.type != 'cuda':
        raise _native.MstError('hard_output needs a GPU tensor; there is no CPU fallback')
    nfeat = x.shape[-1]
    work = x if (x.is_contiguous() and x.dtype == torch.float32) else x.detach().float().contiguous()
    out = torch.empty_like(work)
    _native.check(_native.get().lib.mst_hard_output(_native.ptr(work.detach()), _native.ptr(out), work.numel() // nfeat, nfeat,
                                                    _native.current_stream(x.device)), 'mst_hard_output')
    if work is not x:
        with torch.no_grad():
            x[..., 1] = work[..., 1].to(x.dtype)
    return out
