"""tests/golden/stft_sizes.npz: the reference's spectrogram_torch (vits/mel_processing.py:54-74, reflect pad) at STFT sizes
other than the configs' 2048 / 512 / 2048, run in this container, with the oracle's restatement checked against it.
spectrogram_torch_audio (zero pad) needs torchaudio (absent here): the oracle covers it.

Run:  python tools/make_goldens_stft_sizes.py      (only here; /root/reference does not exist on the GPU box)"""
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

# (importing make_goldens installs the import stubs and puts /root/reference on the path)
from make_goldens import O, close, rmel, rng_tensor, save  # noqa: E402

SIZES = [(1024, 256, 1024), (512, 128, 512), (4096, 1024, 4096), (256, 64, 200), (1024, 256, 800), (2048, 512, 1200),
         (2048, 300, 2048), (64, 16, 64), (1280, 320, 1280), (400, 160, 400), (48, 12, 48)]


def main():
    rng = np.random.default_rng(20241005)
    y = rng_tensor(rng, (2, 5000), 0.25)
    arrs = {"y": y, "sizes": np.array(SIZES)}
    for n_fft, hop, win in SIZES:
        spec = rmel.spectrogram_torch(y, n_fft, 22050, hop, win, center=False)
        close(O.spectrogram(y, n_fft, hop, win, reflect=True), spec, what="spectrogram %d/%d/%d" % (n_fft, hop, win))
        arrs["spec_%d_%d_%d" % (n_fft, hop, win)] = spec
    save("stft_sizes.npz", **arrs)


if __name__ == "__main__":
    main()
