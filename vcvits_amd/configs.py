"""The two hyper-parameter sets of the reference (configs/base.json, configs/48k_base.json of
vtuber-plan/vcvits), restated as dict builders.  Both are 48 kHz / n_fft 2048 / hop 512 /
segment 16384; they differ in widths only."""
import copy

_COMMON = {
    "trainer": {"max_epochs": 20000, "limit_val_batches": 1, "accumulate_grad_batches": 1,
                "default_root_dir": "./logs", "val_check_interval": 1000},
    "train": {"log_interval": 200, "eval_interval": 1000, "seed": 1234, "max_epochs": 20000,
              "learning_rate": 2e-4, "betas": [0.8, 0.99], "eps": 1e-9, "batch_size": 4, "fp16_run": True,
              "lr_decay": 0.999875, "segment_size": 16384, "init_lr_ratio": 1, "warmup_epochs": 0,
              "c_mel": 45, "c_kl": 1},
    "data": {"source_sampling_rate": 16000, "target_sampling_rate": 48000, "filter_length": 2048,
             "hop_length": 512, "win_length": 2048, "n_mel_channels": 256, "mel_fmin": 0.0, "mel_fmax": None,
             "n_speakers": 512, "hubert_channels": 1280, "hubert_downsample": 320, "num_pitch": 512},
    "model": {"num_pitch": 512, "inter_channels": 256, "hidden_channels": 256, "hubert_channels": 1280,
              "filter_channels": 768, "n_heads": 4, "n_layers": 3, "kernel_size": 3, "p_dropout": 0.1,
              "resblock": "1", "resblock_kernel_sizes": [3, 7, 11],
              "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "upsample_rates": [8, 8, 4, 2],
              "upsample_initial_channel": 512, "upsample_kernel_sizes": [16, 16, 4, 4], "n_layers_q": 3,
              "use_spectral_norm": False, "gin_channels": 256},
}


def base():
    """configs/base.json: widths 256, 256 mels, HuBERT x-large features (1280), batch 4; no
    `multi_period_discriminator_periods` key -> class default periods."""
    c = copy.deepcopy(_COMMON)
    c["data"]["max_wav_value"] = 32768.0
    c["data"]["hubert_ckpt"] = c["model"]["hubert_ckpt"] = "checkpoints/hubert_xtralarge_ll60k.pt"
    return c


def base_48k():
    """configs/48k_base.json: widths 128, 128 mels, HuBERT base features (768), batch 16, 12 MPD
    periods."""
    c = copy.deepcopy(_COMMON)
    c["train"]["batch_size"] = 16
    c["data"].update({"n_mel_channels": 128, "hubert_channels": 768})
    c["model"].update({"inter_channels": 128, "hidden_channels": 128, "hubert_channels": 768,
                       "multi_period_discriminator_periods": [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37]})
    c["data"]["hubert_ckpt"] = c["model"]["hubert_ckpt"] = "checkpoints/hubert_base_ls960.pt"
    return c
