"""ctypes binding of libmst_amd.so (include/mst_amd.h) — plumbing only.

The product always loads `music-style-transfer_amd/libmst_amd.so` (hipcc, gfx950) and raises
if it is missing: there is no CPU fallback.  Tests may bind another build of the *same* C ABI
(the hipsim interpreter build of the same .hip sources) by passing an explicit path.
"""
import collections
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(os.path.dirname(_HERE), 'libmst_amd.so')

STAGE_EXTRACT, STAGE_INFO, STAGE_APPLY, STAGE_ALL = 1, 2, 4, 7
LOSS_SAVED = 512
LOSS_KEYS = [
    'total', 'channels_loss_total', 'channels_loss_pitched_total', 'channels_loss_pitched_notes_loss',
    'channels_loss_pitched_velocity_loss', 'channels_loss_pitched_duration_loss',
    'channels_loss_pitched_accidentals_loss', 'channels_loss_unpitched_total',
    'channels_loss_unpitched_notes_loss', 'channels_loss_unpitched_velocity_loss',
    'channels_loss_unpitched_duration_loss', 'song_info_loss_total', 'song_info_loss_instruments_loss',
    'song_info_loss_mode_loss', 'song_info_loss_bpm_loss',
]
N_LOSSES = len(LOSS_KEYS)


class MstError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('C', 'R', 'T', 'beat', 'bar', 'nrf', 'style', 'melody', 'rhythm',
                                         'instr', 'n_instruments', 'has_unpitched', 'clips')]

    def key(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


class PlanOptions(C.Structure):
    _fields_ = [('gemm_tile', C.c_int32), ('no_merge', C.c_int32), ('gemm_run', C.c_int32), ('tile_r0', C.c_int32),
                ('tile_rows', C.c_int32), ('lstm_flavour', C.c_int32), ('dense_flavour', C.c_int32), ('branches', C.c_int32)]


DEV_LSTM_TIMEOUT = 1        # mst_amd.h MST_DEV_LSTM_TIMEOUT


def describe_status(word):
    parts = []
    if word & DEV_LSTM_TIMEOUT:
        parts.append('a granule of the multi-workgroup LSTM exchange did not arrive within 0.2 s (MST_DEV_LSTM_TIMEOUT): the '
                     'launch was not fully resident — something else occupied the GPU next to a batched plan; results since then '
                     'are NaN.  Plan option lstm_flavour=1 (MST_LSTM_FLAVOUR=1) takes the single-workgroup kernels')
    if word & ~DEV_LSTM_TIMEOUT:
        parts.append(f'unknown status bits {word & ~DEV_LSTM_TIMEOUT:#x}')
    return '; '.join(parts)


def options_from_env():
    """Developer switches of the Python binding; the C library itself reads no environment.
    MST_GEMM=mfma|valu forces the 64x64 / 32x32 GEMM tiling, MST_NO_MERGE=1 gives one launch per scheduled member,
    MST_LSTM_FLAVOUR=1 keeps the H = 192 LSTM on one workgroup per sequence (mst_plan_options.lstm_flavour),
    MST_DENSE_FLAVOUR / MST_BRANCHES set mst_plan_options.dense_flavour / .branches."""
    ge = os.environ.get('MST_GEMM')
    if ge not in (None, '', 'mfma', 'valu'):
        raise MstError(f'MST_GEMM={ge!r}: expected mfma or valu')
    return dict(gemm_tile={'mfma': 64, 'valu': 32}.get(ge, 0), no_merge=int(bool(os.environ.get('MST_NO_MERGE'))),
                gemm_run=int(os.environ.get('MST_GEMM_RUN', '0')), lstm_flavour=int(os.environ.get('MST_LSTM_FLAVOUR', '0')),
                dense_flavour=int(os.environ.get('MST_DENSE_FLAVOUR', '0')), branches=int(os.environ.get('MST_BRANCHES', '0')))


_P = C.c_void_p
_SIGS = {
    'mst_param_count': (C.c_int32, [C.POINTER(Dims)]),
    'mst_param_floats': (C.c_int64, [C.POINTER(Dims)]),
    'mst_param_info': (C.c_int32, [C.POINTER(Dims), C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32 * 3)]),
    'mst_widths_supported': (C.c_int32, [C.POINTER(Dims)]),
    'mst_plan_create': (_P, [C.POINTER(Dims), C.POINTER(C.c_int32)]),
    'mst_plan_create_ex': (_P, [C.POINTER(Dims), C.POINTER(PlanOptions), C.POINTER(C.c_int32)]),
    'mst_plan_gemm_tile': (C.c_int32, [_P]),
    'mst_plan_destroy': (None, [_P]),
    'mst_plan_workspace_floats': (C.c_int64, [_P]),
    'mst_plan_tensor': (C.c_int32, [_P, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'mst_plan_launch_count': (C.c_int32, [_P, C.c_int32, C.c_int32]),
    'mst_plan_layout': (C.c_int32, [_P, C.POINTER(C.c_int64 * 4)]),
    'mst_plan_status': (C.c_int32, [_P, _P, C.c_int32, C.POINTER(C.c_int32), _P]),
    'mst_debug_slab_columns_disjoint': (C.c_int32, [C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int32]),
    'mst_forward': (C.c_int32, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    'mst_backward': (C.c_int32, [_P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    'mst_zero_grads': (C.c_int32, [_P, C.c_int32, _P, _P]),
    'mst_plan_zero_floats': (C.c_int64, [_P, C.c_int32]),
    'mst_loss_scratch_floats': (C.c_int64, []),
    'mst_total_loss_fwd': (C.c_int32, [_P, _P, C.c_int64, _P, _P, C.c_int64, _P, _P, C.c_int32, _P, _P, _P, _P,
                                       C.c_int32, _P, _P, _P, _P]),
    'mst_total_loss_bwd': (C.c_int32, [_P, _P, C.c_int64, _P, _P, C.c_int64, _P, _P, C.c_int32, _P, _P, _P, _P,
                                       _P, _P, _P, _P, _P, _P, _P, _P]),
    'mst_train_iteration': (C.c_int32, [_P, _P, _P, _P, _P, _P, _P, _P]),
    'mst_tiled_phase_count': (C.c_int32, [_P]),
    'mst_tiled_phase': (C.c_int32, [_P, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_int32, _P, C.POINTER(C.c_int64 * 8), C.POINTER(C.c_int64 * 8),
                        C.POINTER(C.c_int32)]),
    'mst_adam_step': (C.c_int32, [_P, _P, _P, _P, C.c_int64, _P, C.c_double, C.c_double, C.c_double, C.c_double,
                                  C.c_int32, C.c_double, C.c_int32, _P]),
    'mst_adam_step2': (C.c_int32, [_P, _P, _P, _P, _P, C.c_int64, _P, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.c_int32, C.c_double, C.c_int32, _P]),
    'mst_hard_output': (C.c_int32, [_P, _P, C.c_int64, C.c_int32, _P]),
    'mst_plan_step_count': (C.c_int32, [_P, C.c_int32, C.c_int32]),
    'mst_plan_step_info': (C.c_int32, [_P, C.c_int32, C.c_int32, _P]),
    'mst_plan_step_gemms': (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32]),
    'mst_plan_time_steps': (C.c_int32, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_int32, _P, _P, _P, _P]),
    'mst_audio_plan_create': (_P, [C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_int32)]),
    'mst_audio_plan_destroy': (None, [_P]),
    'mst_audio_plan_info': (C.c_int32, [_P, C.POINTER(C.c_int64 * 6)]),
    'mst_audio_stft': (C.c_int32, [_P, _P, _P, _P, _P]),
    'mst_audio_gram': (C.c_int32, [_P, _P, _P, _P, _P]),
    'mst_audio_style_iteration': (C.c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_double, _P]),
    'mst_version': (C.c_char_p, []),
}


def ptr(t):
    """Raw address of a contiguous fp32 tensor (None -> NULL)."""
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise MstError('mst_amd kernels take contiguous float32 tensors')
    return t.data_ptr()


def check(status, what):
    if status != 0:
        names = {-1: 'MST_ERR_ARG', -2: 'MST_ERR_UNSUPPORTED', -3: 'MST_ERR_LAUNCH', -4: 'MST_ERR_ALLOC'}
        raise MstError(f'{what} failed: {names.get(status, status)}')


class Native:
    def __init__(self, path=None):
        path = path or DEFAULT_LIB
        if not os.path.exists(path):
            raise MstError(
                f'{path} not found: the HIP library has not been built. Run `python -c "import __graft_entry__ as g; '
                f'g.build()"` (hipcc --offload-arch=gfx950). There is no CPU fallback.')
        self.path = path
        self.lib = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(self.lib, name)      # raises AttributeError if a declared symbol is missing
            fn.restype, fn.argtypes = res, args
        # plans (schedule + descriptors + a workspace of 100 MB per clip and up) are cached per (dims, device); training on
        # real songs sees a new (C, R) nearly every iteration, so the cache is a small LRU, not a dict that grows forever
        self._plans = collections.OrderedDict()
        self._plan_cap = int(os.environ.get('MST_PLAN_CACHE', '16'))

    # ---- parameter layout
    def param_table(self, dims):
        n = self.lib.mst_param_count(C.byref(dims))
        if n <= 0:
            raise MstError('bad dims')
        out = []
        buf = C.create_string_buffer(256)
        off, nd, shp = C.c_int64(), C.c_int32(), (C.c_int32 * 3)()
        for i in range(n):
            check(self.lib.mst_param_info(C.byref(dims), i, buf, 256, C.byref(off), C.byref(nd), C.byref(shp)), 'mst_param_info')
            out.append((buf.value.decode(), off.value, tuple(shp[k] for k in range(nd.value))))
        return out

    def param_floats(self, dims):
        return self.lib.mst_param_floats(C.byref(dims))

    def plan(self, dims, device):
        opts = options_from_env()
        key = (dims.key(), str(device), tuple(sorted(opts.items())))
        if key in self._plans:
            self._plans.move_to_end(key)
            return self._plans[key]
        plan = self._plans[key] = Plan(self, dims, device, **opts)
        while len(self._plans) > max(1, self._plan_cap):
            self._plans.popitem(last=False)       # the Plan (and its workspace) dies when its last autograd user lets go
        return plan


def current_stream(device):
    if torch.device(device).type == 'cuda':
        return torch.cuda.current_stream(device).cuda_stream
    return None


class Plan:
    """One mst_plan + its workspace tensor. Tensors returned by `view`/`grad` alias the workspace."""
    WS_POOL_CAP = 4

    def __init__(self, native, dims, device, gemm_tile=None, no_merge=None, gemm_run=None, tile_r0=0, tile_rows=0, lstm_flavour=None, dense_flavour=None, branches=None):
        self.native, self.lib, self.dims, self.device = native, native.lib, dims, torch.device(device)
        env = options_from_env()
        opts = PlanOptions(gemm_tile=env['gemm_tile'] if gemm_tile is None else gemm_tile,
                           no_merge=env['no_merge'] if no_merge is None else int(no_merge),
                           gemm_run=env['gemm_run'] if gemm_run is None else int(gemm_run), tile_r0=int(tile_r0), tile_rows=int(tile_rows),
                           lstm_flavour=env['lstm_flavour'] if lstm_flavour is None else int(lstm_flavour),
                           dense_flavour=env['dense_flavour'] if dense_flavour is None else int(dense_flavour),
                           branches=env['branches'] if branches is None else int(branches))
        st = C.c_int32()
        self.handle = self.lib.mst_plan_create_ex(C.byref(dims), C.byref(opts), C.byref(st))
        if not self.handle:
            check(st.value or -1, 'mst_plan_create_ex')
        self.gemm_tile = self.lib.mst_plan_gemm_tile(self.handle)
        self.branches = bool(opts.branches)
        n = self.lib.mst_plan_workspace_floats(self.handle)
        self.ws = torch.zeros(n, dtype=torch.float32, device=self.device)
        self._free_ws = []
        self._touched = {}        # workspaces launched on since the last health check (check_touched)
        self._slots = {}
        lay = (C.c_int64 * 4)()
        check(self.lib.mst_plan_layout(self.handle, C.byref(lay)), 'mst_plan_layout')
        self.clips, self.clip_stride = int(lay[0]), int(lay[1])

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.mst_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def slot(self, name):
        if name not in self._slots:
            off, goff, numel = C.c_int64(), C.c_int64(), C.c_int64()
            check(self.lib.mst_plan_tensor(self.handle, name.encode(), C.byref(off), C.byref(goff), C.byref(numel)),
                  f'mst_plan_tensor({name})')
            self._slots[name] = (off.value, goff.value, numel.value)
        return self._slots[name]

    def new_ws(self):
        """A fresh workspace for this plan."""
        return torch.zeros_like(self.ws)

    def acquire_ws(self):
        """A workspace for one grad-enabled forward whose backward is pending: taken from the plan's pool (nothing is
        zero-filled on reuse: every slot is written before it is read), allocated only when the pool is empty.  The pool
        holds FREE workspaces only, so one whose backward never runs is simply garbage-collected."""
        return self._free_ws.pop() if self._free_ws else self.new_ws()

    def release_ws(self, ws):
        if ws is not self.ws and len(self._free_ws) < self.WS_POOL_CAP:
            self._free_ws.append(ws)

    def view(self, name, shape=None, ws=None, clip=0):
        off, _, n = self.slot(name)
        off += clip * self.clip_stride
        t = (self.ws if ws is None else ws)[off:off + n]
        return t.view(*shape) if shape is not None else t

    def grad(self, name, shape=None, ws=None, clip=0):
        _, goff, n = self.slot(name)
        goff += clip * self.clip_stride
        t = (self.ws if ws is None else ws)[goff:goff + n]
        return t.view(*shape) if shape is not None else t

    def set_inputs(self, mode=None, bpm=None, instr=None, used=None, bpm_target=None, ws=None, clip=0):
        for name, t in (('mode', mode), ('bpm', bpm), ('instr', instr), ('used_instruments', used), ('bpm_target', bpm_target)):
            if t is not None:
                self.view(name, ws=ws, clip=clip).copy_(torch.as_tensor(t, dtype=torch.float32).reshape(-1), non_blocking=True)

    def status(self, ws=None, clear=True):
        """Device status word of a workspace (0 = healthy; MST_DEV_* bits otherwise).  Synchronises the current stream."""
        word = C.c_int32()
        check(self.lib.mst_plan_status(self.handle, ptr(self.ws if ws is None else ws), int(bool(clear)), C.byref(word),
                                       current_stream(self.device)), 'mst_plan_status')
        return word.value

    def check_status(self, ws=None):
        word = self.status(ws)
        if word:
            raise MstError('device status: ' + describe_status(word))

    def _touch(self, ws):
        if self.branches and self.device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            # torch.cuda.graph's capture_end crashes on a capture that forks into this plan's side streams (ROCm 7.2, torch 2.10)
            raise MstError('a plan made with branches=1 cannot run under torch.cuda.graph capture; capture needs branches=0')
        ws = self.ws if ws is None else ws
        self._touched[id(ws)] = ws
        return ws

    def check_touched(self):
        """check_status of every workspace a launch went to since the last call."""
        touched, self._touched = self._touched, {}
        for ws in touched.values():
            self.check_status(ws)

    def launch_count(self, mask=STAGE_ALL, backward=False):
        return self.lib.mst_plan_launch_count(self.handle, mask, int(backward))

    def forward(self, mask, params, pitched, unpitched, ws=None):
        check(self.lib.mst_forward(self.handle, mask, ptr(params), ptr(self._touch(ws)), ptr(pitched),
                                   ptr(unpitched), current_stream(self.device)), 'mst_forward')

    def backward(self, mask, params, gparams, pitched, unpitched, ws=None):
        check(self.lib.mst_backward(self.handle, mask, ptr(params), ptr(gparams), ptr(self._touch(ws)),
                                    ptr(pitched), ptr(unpitched), current_stream(self.device)), 'mst_backward')

    def zero_grads(self, mask, ws=None):
        check(self.lib.mst_zero_grads(self.handle, mask, ptr(self.ws if ws is None else ws), current_stream(self.device)),
              'mst_zero_grads')

    def time_steps(self, mask, backward, params, gparams, pitched, unpitched, reps=20):
        """[(kind, avg_ms, flops, bytes)] per launch step (HIP events on the current stream)."""
        import numpy as np
        n = self.lib.mst_plan_step_count(self.handle, mask, int(backward))
        ms, kind = np.zeros(n, np.float32), np.zeros(n, np.int32)
        fl, by = np.zeros(n, np.float64), np.zeros(n, np.float64)
        got = self.lib.mst_plan_time_steps(self.handle, mask, int(backward), ptr(params), ptr(gparams), ptr(self.ws),
                                           ptr(pitched), ptr(unpitched), current_stream(self.device), reps,
                                           ms.ctypes.data, kind.ctypes.data, fl.ctypes.data, by.ctypes.data)
        if got != n:
            check(got if got < 0 else -1, 'mst_plan_time_steps')
        return list(zip(kind.tolist(), ms.tolist(), fl.tolist(), by.tolist()))

    def step_gemms(self, mask, backward, step, cap=2048):
        """[(M, N, K, k_splits, fold_rows, workgroups)] of the members (one clip's worth) of GEMM launch step `step`."""
        import numpy as np
        out = np.zeros((cap, 6), np.int32)
        n = self.lib.mst_plan_step_gemms(self.handle, mask, int(backward), step, out.ctypes.data, cap)
        check(n if n < 0 else 0, 'mst_plan_step_gemms')
        return [tuple(r) for r in out[:n].tolist()]

    def tiled_train_iteration(self, params, gparams, pitched, unpitched, losses=None, is_root=True, all_reduce=None):
        """One loop body of a clip whose bars are tiled over ranks (plan made with tile_r0 / tile_rows; `pitched` /
        `unpitched` hold this rank's bars only).  `all_reduce(tensor)` sums a workspace range over the ranks in place —
        torch.distributed.all_reduce by default.  gparams receives this rank's share: all-reduce it before the optimizer step."""
        if all_reduce is None:
            import torch.distributed as dist
            all_reduce = lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM)
        xoff, xlen, nx = (C.c_int64 * 8)(), (C.c_int64 * 8)(), C.c_int32()
        self._touch(None)
        for ph in range(self.lib.mst_tiled_phase_count(self.handle)):
            check(self.lib.mst_tiled_phase(self.handle, ph, ptr(params), ptr(gparams), ptr(self.ws), ptr(pitched), ptr(unpitched),
                                           ptr(losses), int(bool(is_root)), current_stream(self.device), C.byref(xoff), C.byref(xlen),
                                           C.byref(nx)), f'mst_tiled_phase({ph})')
            if nx.value == 1:
                all_reduce(self.ws[xoff[0]:xoff[0] + xlen[0]])
            elif nx.value > 1:
                # the exchanges of one dependency level travel as ONE collective: pack, reduce, unpack
                parts = [self.ws[xoff[q]:xoff[q] + xlen[q]] for q in range(nx.value)]
                total = sum(int(xlen[q]) for q in range(nx.value))
                if getattr(self, '_xstage', None) is None or self._xstage.numel() < total:
                    self._xstage = torch.empty(total, dtype=torch.float32, device=self.device)
                stage = self._xstage[:total]
                torch.cat(parts, out=stage)
                all_reduce(stage)
                at = 0
                for t in parts:
                    t.copy_(stage[at:at + t.numel()])
                    at += t.numel()

    def train_iteration(self, params, gparams, pitched, unpitched, losses=None, ws=None):
        check(self.lib.mst_train_iteration(self.handle, ptr(params), ptr(gparams), ptr(self._touch(ws)),
                                           ptr(pitched), ptr(unpitched), ptr(losses), current_stream(self.device)),
              'mst_train_iteration')


_native = None


def get():
    """The product library (raises MstError when it has not been built)."""
    global _native
    if _native is None:
        _native = Native()
    return _native
