"""the validation summary (vcvits.py:185-245, utils.py:61-69).  The writer calls and the
tensors handed to them are checked; the image is an eyeballing aid (matplotlib-free), not a parity item."""
import numpy as np
import pytest
import torch


class _Writer:
    def __init__(self):
        self.calls = []

    def add_scalar(self, k, v, step):
        self.calls.append(("scalar", k, v, step))

    def add_histogram(self, k, v, step):
        self.calls.append(("histogram", k, v, step))

    def add_image(self, k, v, step, dataformats=None):
        self.calls.append(("image", k, v, step, dataformats))

    def add_audio(self, k, v, step, sr):
        self.calls.append(("audio", k, v, step, sr))


def test_summarize_and_image_cpu():
    from vcvits_amd import utils
    w = _Writer()
    utils.summarize(w, 7, scalars={"a": 1.0}, histograms={"h": np.arange(3)}, images={"i": np.zeros((2, 2, 3), np.uint8)},
                    audios={"x": np.zeros(5)}, audio_sampling_rate=48000)
    assert [c[0] for c in w.calls] == ["scalar", "histogram", "image", "audio"]
    assert w.calls[2][4] == "HWC" and w.calls[3][4] == 48000 and all(c[3] == 7 for c in w.calls)
    mel = np.linspace(-11.5, 2.0, 80 * 37, dtype=np.float32).reshape(80, 37)
    img = utils.plot_spectrogram_to_numpy(mel)
    assert img.shape == (200, 1000, 3) and img.dtype == np.uint8
    # origin='lower': the lowest channels (smallest values here) are at the bottom rows
    assert tuple(img[-1, 0]) == (68, 1, 84) and tuple(img[0, -1]) == (253, 231, 37)
    flat = utils.plot_spectrogram_to_numpy(np.full((4, 4), 3.0, np.float32))
    assert (flat == flat[0, 0]).all()
    with pytest.raises(ValueError):
        utils.plot_spectrogram_to_numpy(np.zeros((0, 4), np.float32))


@pytest.mark.gpu
def test_validation_step_writes_reference_summary(gpu):
    import types
    from vcvits_amd import synthetic
    from vcvits_amd.light.vcvits import VCVITS
    from test_training_step_gpu import small_cfg
    torch.manual_seed(3)
    cfg = small_cfg()
    module = VCVITS(**cfg).to(gpu)
    batch = {k: v.to(gpu) for k, v in synthetic.full_batch(2, 24, t_y=40, t_x=22, seed=5).items()}
    batch["sid"] = batch["sid"] % 8
    ret_plain = module.validation_step(batch, 0)          # no logger: nothing to write
    w = _Writer()
    module.logger = types.SimpleNamespace(experiment=w)
    module.global_step = 11
    y_hat, y_hat_lengths, mel, y_hat_mel = module.validation_step(batch, 0)
    assert len(ret_plain) == 4 and mel.shape[0] == 1 and mel.shape[1] == cfg["data"]["n_mel_channels"]
    kinds = [(c[0], c[1]) for c in w.calls]
    assert kinds == [("image", "gen/mel"), ("image", "gt/mel"), ("audio", "gen/audio"), ("audio", "gt/audio")]
    assert all(c[3] == 11 for c in w.calls)
    for c in w.calls[:2]:
        assert c[2].shape == (200, 1000, 3) and c[2].dtype == np.uint8 and c[4] == "HWC"
    gen, gt = w.calls[2], w.calls[3]
    assert gen[4] == gt[4] == cfg["data"]["target_sampling_rate"]
    assert gen[2].shape == (1, int(y_hat_lengths[0])) and gt[2].shape == (1, int(batch["y_wav_lengths"][0]))
    assert module.net_g.training                          # eval() only for the duration of the step
