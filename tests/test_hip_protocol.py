"""The reference's customisation point (README "implement stft / separate / istft"; css/css.py:131,199): the drop-in
`separate_and_stitch` takes ANY object with the separator protocol.  Here a toy torch module is the mask estimator; the
HIP stages do the rest, and the result is held against the oracle driven by the very masks the toy returned (decisions
exact, waveforms <= 1e-4 relative RMS).  Needs an MI355X."""
import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu


def _toy(torch, S, rotate):
    class Toy(torch.nn.Module):
        """masks from the magnitude of microphone 0 and smooth patterns over (bin, frame); call i rotates the speaker
        channels by i mod S (non-trivial stitching permutations)"""

        def __init__(self):
            super().__init__()
            self.calls, self.seen, self.ndims = 0, [], []

        def stft(self, s):      # conformer_wrapper.py:106-129: 512-point periodic Hann, hop 256, no padding
            x = s if s.ndim == 3 else s[..., None]
            w = torch.hann_window(512, periodic=True, device=x.device)
            X = torch.stft(x.permute(0, 2, 1).reshape(-1, x.shape[1]), 512, 256, window=w, center=False, return_complex=True)
            X = X.reshape(x.shape[0], x.shape[2], 257, -1).permute(0, 2, 3, 1)
            return X if s.ndim == 3 else X[..., 0]

        def separate(self, stft):
            self.ndims.append(stft.ndim)
            x = stft if stft.ndim == 4 else stft[..., None]
            mag = x[..., 0].abs()                                     # [1, F, T]
            F, T = mag.shape[1], mag.shape[2]
            f = torch.arange(F, device=mag.device, dtype=torch.float32)[:, None]
            t = torch.arange(T, device=mag.device, dtype=torch.float32)[None, :]
            logits = [2.5 * torch.sin(0.045 * (k + 1) * f + 0.21 * (k + 2) * t + 1.3 * k) + 0.3 * torch.log10(mag[0] + 1e-3)
                      for k in range(S + 1)]
            m = torch.sigmoid(torch.stack(logits, dim=-1))[None]     # [1, F, T, S + 1]
            spk = m[..., :S]
            if rotate:
                spk = torch.roll(spk, self.calls % S, dims=-1)
            self.calls += 1
            self.seen.append((spk[0].cpu().numpy().copy(), m[0, ..., S:].cpu().numpy().copy()))
            return {'spk_masks': spk, 'noise_masks': m[..., S:]}

        def istft(self, stft):   # (the HIP stages synthesise; kept for the protocol's sake)
            raise AssertionError("not called by the drop-in driver")

    return Toy().eval()


@pytest.mark.parametrize("channels,rotate", [(7, True), (7, False), (1, True)])
def test_foreign_separator_masks_through_the_hip_stages(channels, rotate):
    import torch
    import css_oracle as O
    CSS = pkg("css")
    mix = pkg("synth").synth_meeting(13.7, 7, seed=42)[:, :, :channels]
    toy = _toy(torch, 3, rotate)
    cfg = CSS.CssCfg(activity_th=0.45, show_progressbar=False)
    wavs, side = CSS.separate_and_stitch(mix, toy, 16000, "cuda:0", cfg)
    nseg = toy.calls
    assert nseg == len(toy.seen) and nseg >= 8 and len(wavs) == 3
    assert set(toy.ndims) == {4}                                # [1, F, T, C] as css.py:199 passes it, C == 1 included
    ow, oside = O.separate_and_stitch(mix, None, 16000, O.OracleCssCfg(activity_th=0.45),
                                      separate_fn=lambda i, seg: toy.seen[i], mvdr_cplx=np.complex128)
    assert side['segment_frames'] == 186
    assert np.array_equal(side['activity_b'].numpy(), oside['activity_b'])
    assert np.array_equal(side['activity_final'].numpy(), oside['activity_final'])
    if rotate:
        assert len({tuple(p) for p in oside['perms']}) > 1      # the rotation really forces permutations
    assert float(np.abs(side['mask_stitched'].numpy() - oside['mask_stitched']).max()) < 2e-6
    for k in range(3):
        err = float(np.sqrt(np.mean((wavs[k] - ow[k]) ** 2)) / np.sqrt(np.mean(ow[k] ** 2)))
        assert err < 1e-4, (k, err)


def test_a_separator_with_another_transform_is_rejected():
    import torch
    CSS = pkg("css")
    toy = _toy(torch, 3, False)
    toy.stft = lambda s: torch.zeros((1, 257, 4, 7), dtype=torch.complex64)     # not the stages' transform
    mix = pkg("synth").synth_meeting(4.0, 7, seed=1)
    with pytest.raises(ValueError):
        CSS.separate_and_stitch(mix, toy, 16000, "cuda:0", CSS.CssCfg(show_progressbar=False))
