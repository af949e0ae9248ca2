"""AUDIO EXTENSION — not part of the reference's surface (the reference is a symbolic model with no audio path,
latex/music-style-transfer.tex:79-80).  Thin plumbing over the mst_audio_* entry points of include/mst_amd.h: an STFT(1024/256 or
2048/512, Hann, reflect-centred) featuriser, a feature-Gram on the f32 matrix cores and one spectrogram style-transfer optimisation
iteration (Gram loss + gradient + Adam).  GPU only: like the rest of the package there is no CPU fallback."""
import ctypes as C

import torch

from style import _native


class AudioPlan:
    def __init__(self, n_samples, n_fft=1024, hop=256, device='cuda:0', native=None):
        self.native = native or _native.get()
        self.lib, self.device = self.native.lib, torch.device(device)
        st = C.c_int32()
        self.handle = self.lib.mst_audio_plan_create(n_fft, hop, n_samples, C.byref(st))
        if not self.handle:
            _native.check(st.value or -1, 'mst_audio_plan_create')
        info = (C.c_int64 * 6)()
        _native.check(self.lib.mst_audio_plan_info(self.handle, C.byref(info)), 'mst_audio_plan_info')
        self.frames, self.bins, self.ld, ws_floats, self.splits, self.tiles = (int(v) for v in info)
        self.n_samples, self.n_fft, self.hop = n_samples, n_fft, hop
        self.ws = torch.zeros(ws_floats, dtype=torch.float32, device=self.device)

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.mst_audio_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def stft(self, audio, want_spec=True, want_mag=True):
        """audio: (n_samples,) float32 on the device -> (spec (frames, bins) complex64 | None, mag (frames, ld) | None)."""
        if audio.numel() != self.n_samples:
            raise _native.MstError('audio length differs from the plan')
        spec = torch.empty(self.frames, self.bins, 2, dtype=torch.float32, device=self.device) if want_spec else None
        mag = torch.empty(self.frames, self.ld, dtype=torch.float32, device=self.device) if want_mag else None
        _native.check(self.lib.mst_audio_stft(self.handle, _native.ptr(audio), _native.ptr(spec), _native.ptr(mag),
                                              _native.current_stream(self.device)), 'mst_audio_stft')
        return (torch.view_as_complex(spec) if spec is not None else None), mag

    def gram(self, feat):
        g = torch.empty(self.ld, self.ld, dtype=torch.float32, device=self.device)
        _native.check(self.lib.mst_audio_gram(self.handle, _native.ptr(feat), _native.ptr(g), _native.ptr(self.ws),
                                              _native.current_stream(self.device)), 'mst_audio_gram')
        return g

    def optimizer_state(self):
        z = lambda: torch.zeros(self.frames, self.ld, dtype=torch.float32, device=self.device)
        return dict(grad=z(), exp_avg=z(), exp_avg_sq=z(), state=torch.zeros(4, dtype=torch.float32, device=self.device),
                    loss=torch.zeros(1, dtype=torch.float32, device=self.device))

    def style_iteration(self, x, gram_style, opt, lr=1e-2):
        P = _native.ptr
        _native.check(self.lib.mst_audio_style_iteration(self.handle, P(x), P(gram_style), P(opt['grad']), P(opt['exp_avg']),
                                                         P(opt['exp_avg_sq']), P(opt['state']), P(self.ws), P(opt['loss']), lr,
                                                         _native.current_stream(self.device)), 'mst_audio_style_iteration')
        return opt['loss']
