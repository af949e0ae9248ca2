"""TEST INFRASTRUCTURE — CPU oracle of the AUDIO EXTENSION (music-style-transfer_amd/csrc/audio.hip).  PARITY UNPINNED: the
reference (marcinp7/music-style-transfer) has no audio path (latex/music-style-transfer.tex:79-80; requirements.txt:1-8), so there
is nothing of it to restate or to generate fixtures from.  The oracle is build-defined, as SURVEY.md 8(f4) prescribes: torch's
own STFT, matmul, autograd and Adam on the CPU.  Only tests/, __graft_entry__.smoke() and bench.py's baseline legs import this."""
import torch


def stft(audio, n_fft=1024, hop=256):
    """(frames, bins) complex64: periodic Hann window, centre-padded by reflection (torch.stft's defaults), frames = 1 + n // hop."""
    spec = torch.stft(audio, n_fft, hop_length=hop, window=torch.hann_window(n_fft), center=True, pad_mode='reflect',
                      return_complex=True)
    return spec.transpose(0, 1).contiguous()


def gram(feat):
    """Feature Gram of a (frames, bins) real matrix: feat^T feat / frames."""
    return feat.transpose(0, 1) @ feat / feat.shape[0]


def style_loss(x, gram_style):
    d = gram(x) - gram_style
    return (d * d).sum()


def style_iterations(x0, gram_style, n, lr):
    """n Adam iterations on x; returns (x_n, [loss before each step])."""
    x = x0.clone().requires_grad_(True)
    opt = torch.optim.Adam([x], lr=lr)
    losses = []
    for _ in range(n):
        opt.zero_grad()
        loss = style_loss(x, gram_style)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return x.detach(), losses
