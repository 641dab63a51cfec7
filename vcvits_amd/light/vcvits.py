"""VCVITS training module -- the step semantics of vits/light/vcvits.py:28-283 of the reference.

The class IS a `pytorch_lightning.LightningModule` wherever that package imports (the reference's is:
vits/light/vcvits.py:14,28, handed to `Trainer.fit` by train.py:85,110-113, which type-checks it) and a plain
`nn.Module` otherwise (this image has no Lightning).  Either way it carries its own loop: Lightning-1.x's
"for optimizer_idx in (0, 1): training_step -> backward -> step" is restated in `fit_batch`, which is what the
benchmarks, the tests and the HIP-graph replay drive; under a Lightning Trainer the hooks below (`training_step(batch,
batch_idx, optimizer_idx)`, `configure_optimizers` -> two `torch.optim.Optimizer`s + two schedulers,
`validation_step`, `on_load_checkpoint`) are what it calls.

Kept: attribute names (net_g, net_period_d, net_scale_d, hparams), `training_step(batch,
batch_idx, optimizer_idx)`, `validation_step`, `configure_optimizers`, `on_load_checkpoint`, the
batch-dict schema of vits/data/collate.py:177-187, the loss composition (:113-117) and the
reference's quirks (zero-padded STFT for training targets, mel target = slice of the
full-utterance mel, MPD default periods when the config omits them).
Batch schema: the reference's (`x_wav_values` / `x_wav_lengths`, what vits/data/collate.py emits): the source
waveform goes through `audio_pipeline` (vcvits.py:61-62) and the content encoder's pluggable HuBERT feature
extractor (`set_feature_extractor`; the fairseq model itself is third-party and absent offline).  Batches that
carry precomputed content features `x_hubert_features_values` [B, hubert, T'] instead (every benchmark
configuration) skip both."""
import itertools
import logging
from typing import Any, Dict

import torch
from torch import nn

from .. import commons, ops
from ..hparams import HParams
from ..losses import discriminator_loss, feature_loss, generator_loss, kl_loss
from ..mel_processing import mel_spectrogram_torch, spec_to_mel_torch, spectrogram_torch_audio
from ..model.discriminators.multi_period_discriminator import MultiPeriodDiscriminator
from ..model.discriminators.multi_scale_discriminator import MultiScaleDiscriminator
from ..model.generator import Generator
from ..model.synthesizers.synthesizer_svc import SynthesizerSVC
from .optim import ExponentialLR, FlatAdamW

DEFAULT_PERIODS = [2, 3, 5, 7, 11, 17, 23, 37]  # multi_period_discriminator.py:10


def _hp(v):
    return HParams(**v) if isinstance(v, dict) else v


try:  # the reference's base class, where it exists (vits/light/vcvits.py:14)
    import pytorch_lightning as _pl
    _Base, HAS_LIGHTNING = _pl.LightningModule, True
except Exception:  # noqa: BLE001 -- not installed (this image) or broken: the module carries its own loop anyway
    _Base, HAS_LIGHTNING = nn.Module, False


from ..model.discriminators._pair import join_streams  # noqa: E402

class VCVITS(_Base):
    def __init__(self, **kwargs):
        super().__init__()
        hp = HParams(**{k: (v.to_dict() if isinstance(v, HParams) else v) for k, v in kwargs.items()})
        if HAS_LIGHTNING:
            # vcvits.py:31: `self.save_hyperparameters(*[k for k in kwargs])` -- Lightning collects them from this frame's
            # `kwargs`; nested dicts become HParams first so `self.hparams.data.hop_length` reads as in the reference
            kwargs = {k: hp[k] for k in kwargs}
            self.save_hyperparameters(*[k for k in kwargs])
        else:
            self.hparams = hp
        hp = self.hparams
        self.net_g = self._build_generator()
        periods = hp.model.get("multi_period_discriminator_periods", None) or DEFAULT_PERIODS
        self.net_period_d = MultiPeriodDiscriminator(periods=list(periods),
                                                     use_spectral_norm=hp.model.use_spectral_norm)
        self.net_scale_d = MultiScaleDiscriminator(hp.model.use_spectral_norm)
        from ..model.pipeline import SpeechConversionAudioPipeline
        # STFT -> iSTFT pass over the 16 kHz source waveform (vcvits.py:45-52,62); its output feeds the content
        # encoder's feature extractor when the batch carries `x_wav_values`
        self.audio_pipeline = SpeechConversionAudioPipeline(sr=hp.data.source_sampling_rate, n_fft=hp.data.filter_length,
                                                            n_mel=hp.data.n_mel_channels, win_length=hp.data.win_length,
                                                            hop_length=hp.data.hop_length)
        self._own_epoch = 0
        self._own_step = 0
        self.logged = {}
        self.optim_g = self.optim_d = None
        self.scheduler_g = self.scheduler_d = None

    # `current_epoch` / `global_step`: the attached Trainer's when there is one (LightningModule's read-only properties),
    # this module's own counters otherwise (fit_batch / on_epoch_end advance them; checkpoints restore them)
    def _trainer_or_none(self):
        return getattr(self, "_trainer", None) if HAS_LIGHTNING else None

    @property
    def current_epoch(self):
        t = self._trainer_or_none()
        return t.current_epoch if t is not None else self._own_epoch

    @current_epoch.setter
    def current_epoch(self, v):
        self._own_epoch = int(v)

    @property
    def global_step(self):
        t = self._trainer_or_none()
        return t.global_step if t is not None else self._own_step

    @global_step.setter
    def global_step(self, v):
        self._own_step = int(v)

    def set_feature_extractor(self, extractor):
        """Plug the frozen HuBERT (or any stand-in with its `extract_features` contract) into the content encoder."""
        self.net_g.enc_p.set_feature_extractor(extractor)

    def _source(self, batch):
        """(x, x_lengths) for net_g from either batch schema.  Reference schema: `x_wav_values` [B, 1, T] ->
        audio_pipeline under no_grad (vcvits.py:61-62) -> content encoder (which runs the feature extractor);
        lengths stay SAMPLE counts, as the reference passes them (content_encoder.py:66 quirk)."""
        if "x_wav_values" in batch:
            with torch.no_grad():
                return self.audio_pipeline(batch["x_wav_values"]), batch["x_wav_lengths"]
        if "x_hubert_features_values" in batch:
            return batch["x_hubert_features_values"], batch["x_hubert_features_lengths"]
        raise KeyError("batch has neither the reference's source-waveform keys (x_wav_values / x_wav_lengths) nor "
                       "precomputed content features (x_hubert_features_values / x_hubert_features_lengths); "
                       "keys: %s" % sorted(batch))

    def _build_generator(self):
        hp = self.hparams
        return SynthesizerSVC(hp.data.filter_length // 2 + 1, hp.train.segment_size // hp.data.hop_length,
                              n_speakers=hp.data.n_speakers, **hp.model)

    # ------------------------------------------------------------------------------------------
    def _spec_mel(self, wav):
        d = self.hparams.data
        spec = spectrogram_torch_audio(wav, d.filter_length, d.target_sampling_rate, d.hop_length, d.win_length,
                                       center=False)
        mel = spec_to_mel_torch(spec, d.filter_length, d.n_mel_channels, d.target_sampling_rate, d.mel_fmin,
                                d.mel_fmax)
        return spec, mel

    def _generator_pass(self, batch, decoder_only=False):
        """vcvits.py:55-82: targets (no grad), net_g forward, waveform slice.  Returns
        (y_hat, y, y_mel_slice, kl_args or None).  decoder_only (the discriminator step's no-grad pass, which keeps y_hat
        alone): the generator computes only what y_hat depends on (SynthesizerSVC.forward)."""
        d, t = self.hparams.data, self.hparams.train
        speakers = batch.get("sid", None)
        # (decoder_only: the content encoder does not run, so neither does the source pipeline that feeds it)
        x_feat, x_lengths = (None, None) if decoder_only else self._source(batch)
        x_pitch, x_pitch_lengths = batch["x_pitch_values"], batch["x_pitch_lengths"]
        y_wav, y_wav_lengths = batch["y_wav_values"], batch["y_wav_lengths"]
        with torch.no_grad():
            if decoder_only:  # (the target mel is a loss operand of the generator step only)
                y_mel = None
                y_spec = spectrogram_torch_audio(y_wav.squeeze(1), d.filter_length, d.target_sampling_rate, d.hop_length,
                                                 d.win_length, center=False)
            else:
                y_spec, y_mel = self._spec_mel(y_wav.squeeze(1))
            y_spec_lengths = (y_wav_lengths / d.hop_length).long()
        y_hat, ids_slice, z_slice, x_mask, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q) = \
            self.net_g(x_feat, x_lengths, x_pitch, x_pitch_lengths, y_spec, y_spec_lengths, sid=speakers,
                       noise=batch.get("noise", None), ids_slice=batch.get("ids_slice", None), decoder_only=decoder_only,
                       raw_sizes=batch.get("bucket_raw_sizes", None))
        with torch.no_grad():
            y = ops.slice_segments(y_wav, ids_slice, t.segment_size, d.hop_length)
            y_mel_slice = None if decoder_only else commons.slice_segments(y_mel, ids_slice, t.segment_size // d.hop_length)
        return y_hat, y, y_mel_slice, (z_p, logs_q, m_p, logs_p, z_mask)

    def _nograd_generator_pass(self, batch):
        """(y_hat, y) of the discriminator step's generator pass: a HIP-graph replay once the batch shapes repeat
        (light/graphed.py), the eager pass otherwise."""
        g = self.__dict__.get("_g_graph")
        if g is None:
            from .graphed import GraphedNoGrad
            # (with the dropout trace of tests/test_dropout_step_gpu.py on, the pass stays whole: the trace lists its draws)
            g = self.__dict__["_g_graph"] = GraphedNoGrad(
                lambda b: self._generator_pass(b, decoder_only=ops.DROPOUT_TRACE[0] is None)[:2])
        if ops.DROPOUT_TRACE[0] is None:
            # the decoder-only pass reads the target waveform, the speaker ids and the two injected draws: only those are
            # copied into the graph's static inputs (and only their shapes key it)
            batch = {k: batch[k] for k in ("y_wav_values", "y_wav_lengths", "sid", "noise", "ids_slice", "z_slice",
                                           "x_pitch_values", "x_pitch_lengths") if k in batch}
        return g(batch, extra=(self.training, ops.compute_dtype(), ops._USE_X3[0], ops.bf16_activations()))

    def drop_graphs(self):
        """Forget every recorded HIP graph of this module: they bake parameter addresses (call after anything that moves
        parameter storage -- a rebuilt optimizer, `.to()`, a state_dict that swaps a layer's form)."""
        for name in ("_g_graph", "_batch_graph"):
            g = self.__dict__.pop(name, None)
            if g is not None:
                g.drop()

    def _apply(self, fn, *args, **kwargs):
        self.drop_graphs()
        ops.GRAPH_EPOCH[0] += 1
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.drop_graphs()
        ops.GRAPH_EPOCH[0] += 1
        out = super().load_state_dict(*args, **kwargs)
        ops.invalidate_weights()
        return out

    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int, optimizer_idx: int):
        t = self.hparams.train
        if optimizer_idx == 0:
            y_hat, y, y_mel_slice, kl_args = self._generator_pass(batch)
            y_dp_hat_r, y_dp_hat_g, fmap_p_r, fmap_p_g = self.net_period_d(y, y_hat)
            loss_p_fm = feature_loss(fmap_p_r, fmap_p_g)
            loss_p_gen, _ = generator_loss(y_dp_hat_g)
            y_ds_hat_r, y_ds_hat_g, fmap_s_r, fmap_s_g = self.net_scale_d(y, y_hat)
            loss_s_fm = feature_loss(fmap_s_r, fmap_s_g)
            loss_s_gen, _ = generator_loss(y_ds_hat_g)
            _, y_mel_hat = self._spec_mel(y_hat)  # [B,1,n_mel,32]
            y_mel_hat = y_mel_hat.squeeze(1)
            loss_mel = ops.l1_mean_sum([y_mel_hat], [y_mel_slice], weight=float(t.c_mel))
            loss_gen_all = (loss_s_gen + loss_s_fm) + (loss_p_gen + loss_p_fm) + loss_mel
            self.logged = {"loss/g/p_fm": loss_p_fm, "loss/g/s_fm": loss_s_fm, "loss/g/p_gen": loss_p_gen,
                           "loss/g/s_gen": loss_s_gen, "loss/g/loss_mel": loss_mel}
            if kl_args is not None:
                loss_kl = kl_loss(*kl_args) * t.c_kl
                loss_gen_all = loss_gen_all + loss_kl
                self.logged["loss/g/loss_kl"] = loss_kl
            self.logged["loss/g/total"] = loss_gen_all
            return loss_gen_all
        if optimizer_idx == 1:
            with torch.no_grad():  # only y_hat.detach() is consumed (vcvits.py:153,157)
                y_hat, y = self._nograd_generator_pass(batch)
            y_dp_hat_r, y_dp_hat_g, _, _ = self.net_period_d(y, y_hat.detach())
            loss_disc_p, r_p, g_p = discriminator_loss(y_dp_hat_r, y_dp_hat_g)
            y_ds_hat_r, y_ds_hat_g, _, _ = self.net_scale_d(y, y_hat.detach())
            loss_disc_s, r_s, g_s = discriminator_loss(y_ds_hat_r, y_ds_hat_g)
            loss_disc_all = loss_disc_p + loss_disc_s
            self.logged = {"loss/d/total": loss_disc_all}
            return loss_disc_all
        raise ValueError("optimizer_idx must be 0 (generator) or 1 (discriminators)")

    def validation_step(self, batch, batch_idx):
        """vcvits.py:185-245: returns (y_hat, y_hat_lengths, mel, y_hat_mel); when a logger with a TensorBoard-style
        `.experiment` writer is attached (`self.logger`), also writes the reference's summary -- gen/mel, gt/mel
        images and gen/audio, gt/audio clips of the first utterance -- through utils.summarize."""
        d = self.hparams.data
        self.net_g.eval()
        with torch.no_grad():
            y_spec, mel = self._spec_mel(batch["y_wav_values"].squeeze(1)[:1])
            # target frames per source SAMPLE (vcvits.py:206, data.infer_length_scale); a feature batch carries
            # lengths in feature frames = source samples / hubert_downsample (content_encoder.py:54-56)
            from ..data.audio import infer_length_scale
            from ..model.encoders.content_encoder import HUBERT_DOWNSAMPLE
            len_scale = infer_length_scale(d)
            if "x_wav_values" not in batch:
                len_scale *= d.get("hubert_downsample", HUBERT_DOWNSAMPLE)
            x_src, x_src_lengths = self._source(batch)
            y_hat, mask, _ = self.net_g.infer(x_src, x_src_lengths,
                                              batch["x_pitch_values"], batch["x_pitch_lengths"],
                                              sid=batch.get("sid", None), length_scale=len_scale, max_len=1000)
            y_hat_lengths = mask.sum([1, 2]).long() * d.hop_length
            y_hat_mel = mel_spectrogram_torch(y_hat.squeeze(1).float(), d.filter_length, d.n_mel_channels,
                                              d.target_sampling_rate, d.hop_length, d.win_length, d.mel_fmin,
                                              d.mel_fmax)
        self.net_g.train()
        ops.check_indices()  # (validation already reads results back: the out-of-range-index count rides along)
        writer = getattr(getattr(self, "logger", None), "experiment", None)
        if writer is not None:
            from .. import utils
            y_wav, y_len = batch["y_wav_values"], batch["y_wav_lengths"]
            utils.summarize(writer=writer, global_step=getattr(self, "global_step", 0),
                            images={"gen/mel": utils.plot_spectrogram_to_numpy(y_hat_mel[0].cpu().numpy()),
                                    "gt/mel": utils.plot_spectrogram_to_numpy(mel[0].cpu().numpy())},
                            audios={"gen/audio": y_hat[0, :, :int(y_hat_lengths[0])],
                                    "gt/audio": y_wav[0, :, :int(y_len[0])]},
                            audio_sampling_rate=d.target_sampling_rate)
        return y_hat, y_hat_lengths, mel, y_hat_mel

    # ------------------------------------------------------------------------------------------
    def generator_parameters(self):
        return self.net_g.parameters()

    def train_stream(self):
        """The module's own HIP stream: `fit_batch` runs eager batches, graph captures and replays on it, and
        `configure_optimizers` registers the gradient hooks under it (light/graphed.py: ONE stream).  None on the CPU."""
        if not torch.cuda.is_available() or not next(self.parameters()).is_cuda:
            return None
        s = self.__dict__.get("_stream")
        if s is None:
            s = self.__dict__["_stream"] = torch.cuda.Stream(device=next(self.parameters()).device)
        return s

    def configure_optimizers(self, process_group=None):
        s = self.train_stream()
        if s is None:
            return self._configure_optimizers(process_group)
        s.wait_stream(torch.cuda.current_stream(s.device))
        with torch.cuda.stream(s):
            out = self._configure_optimizers(process_group)
        torch.cuda.current_stream(s.device).wait_stream(s)
        return out

    def _configure_optimizers(self, process_group=None):
        t = self.hparams.train
        self.drop_graphs()  # recorded graphs bake the addresses of the flat buffers the old optimizers owned
        for old in (self.optim_g, self.optim_d):
            if old is not None:
                old.close()  # hooks / gradient sinks of a replaced optimizer must not outlive it
        # dropout stream: tied to torch's seed, offset per data-parallel rank (independent masks per rank), saved
        # in checkpoints
        import torch.distributed as dist
        rank = dist.get_rank(process_group) if dist.is_available() and dist.is_initialized() else 0
        ops.manual_seed(torch.initial_seed() + 7919 * rank)
        self.optim_g = FlatAdamW(self.generator_parameters(), t.learning_rate, betas=t.betas, eps=t.eps,
                                 process_group=process_group)
        self.optim_d = FlatAdamW(itertools.chain(self.net_period_d.parameters(), self.net_scale_d.parameters()),
                                 t.learning_rate, betas=t.betas, eps=t.eps, process_group=process_group)
        # vcvits.py:258-261: ExponentialLR per optimizer with last_epoch re-seated to current_epoch - 1 (the rate
        # itself starts at learning_rate; a resume restores it with the optimizer state)
        # (`train.lr_reference_stack` in the hparams selects the scheduler semantics: see ExponentialLR)
        ref_stack = t.get("lr_reference_stack", None) if hasattr(t, "get") else getattr(t, "lr_reference_stack", None)
        self.scheduler_g = ExponentialLR(self.optim_g, gamma=t.lr_decay, reference_stack=ref_stack)
        self.scheduler_g.last_epoch = self.current_epoch - 1
        self.scheduler_d = ExponentialLR(self.optim_d, gamma=t.lr_decay, reference_stack=ref_stack)
        self.scheduler_d.last_epoch = self.current_epoch - 1
        return [self.optim_g, self.optim_d], [self.scheduler_g, self.scheduler_d]

    def _toggle(self, idx):
        """Lightning-1.x toggle_optimizer: only the active optimizer's parameters take gradients."""
        g_on = idx == 0
        for p in self.optim_g.params:
            p.requires_grad_(g_on)
        for p in self.optim_d.params:
            p.requires_grad_(not g_on)

    def fit_batch(self, batch, batch_idx=0, after_backward=None):
        """One batch of the reference's loop: generator step, then discriminator step.
        `after_backward(optimizer_idx, optimizer)` is an optional probe called before each step.
        Runs on the module's own stream (train_stream), ordered after the caller's current stream on entry and before it on
        exit -- to the caller it behaves as if it ran on the current stream."""
        if self.optim_g is None:
            self.configure_optimizers()
        s = self.train_stream()
        if s is None:
            return self._fit_batch(batch, batch_idx, after_backward)
        cur = torch.cuda.current_stream(s.device)
        if cur.cuda_stream == s.cuda_stream:
            return self._fit_batch(batch, batch_idx, after_backward)
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            out = self._fit_batch(batch, batch_idx, after_backward)
        cur.wait_stream(s)
        return out

    def _bucketed(self, batch):
        """hparams.train.length_bucket_frames = N > 0: the batch's padded lengths rounded up to multiples of N frames
        (data/collate.py: bucket_batch) so that variable-length batches hit recorded graphs; 0 / absent: as collated."""
        t = self.hparams.train
        n = (t.get("length_bucket_frames", 0) if hasattr(t, "get") else getattr(t, "length_bucket_frames", 0)) or 0
        if n <= 0 or "y_wav_lengths" not in batch or "bucket_raw_sizes" in batch:  # (already bucketed by the collate)
            return batch
        from ..data.collate import bucket_batch, bucket_multiples
        return bucket_batch(batch, bucket_multiples(self.hparams.data.hop_length, int(n)), self.hparams.data.hop_length)

    def _fit_batch(self, batch, batch_idx=0, after_backward=None):
        out = self._fit_batch_inner(batch, batch_idx, after_backward)
        if ops.CHECK_INDICES_EVERY_BATCH[0]:  # (VCVITS_CHECK_INDICES=1: a device sync per batch; default: at the check points)
            ops.check_indices()
        return out

    def _fit_batch_inner(self, batch, batch_idx=0, after_backward=None):
        batch = self._bucketed(batch)
        if after_backward is None:
            # the whole batch -- both passes and their AdamW steps -- replayed from ONE HIP graph once the batch shapes
            # repeat (light/graphed.py); None: run it eagerly (shapes still new, profiler active, graphs off)
            bg = self.__dict__.get("_batch_graph")
            if bg is None:
                from .graphed import GraphedBatch
                bg = self.__dict__["_batch_graph"] = GraphedBatch(self)
            out = bg.run(batch, extra=(ops.compute_dtype(), ops._USE_X3[0], ops._USE_X3_WGRAD[0], ops._USE_PK[0],
                                       ops.bf16_activations(), ops._DETERMINISTIC[0]))
            if out is not None:
                self._own_step += 1
                return dict(out)
        out = {}
        for idx, opt in ((0, self.optim_g), (1, self.optim_d)):
            self._toggle(idx)
            opt.zero_grad()
            loss = self.training_step(batch, batch_idx, idx)
            loss.backward()
            join_streams()  # (VCVITS_STREAMS > 1: the sub-discriminators' gradient kernels ran on side streams)
            if after_backward is not None:
                opt.finish_grad_sync()
                after_backward(idx, opt)
            opt.step()
            out["g" if idx == 0 else "d"] = loss.detach()
        for p in itertools.chain(self.optim_g.params, self.optim_d.params):
            p.requires_grad_(True)
        self._own_step += 1
        return out

    def on_epoch_end(self):
        """Lightning steps both epoch-interval schedulers at the end of every training epoch."""
        self._own_epoch += 1
        ops.check_indices()
        self.scheduler_g.step()
        self.scheduler_d.step()

    def on_load_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
        """vcvits.py:265-282: shape-mismatched tensors are replaced by the fresh ones, unknown keys
        are dropped, and the optimizer state is discarded if anything changed."""
        state_dict = checkpoint["state_dict"]
        model_state_dict = self.state_dict()
        is_changed = False
        for k in list(state_dict):
            if k in model_state_dict:
                if state_dict[k].shape != model_state_dict[k].shape:
                    logging.info(f"Skip loading parameter: {k}, required shape: {model_state_dict[k].shape}, "
                                 f"loaded shape: {state_dict[k].shape}")
                    state_dict[k] = model_state_dict[k]
                    is_changed = True
            else:
                logging.info(f"Dropping parameter {k}")
                is_changed = True
        if is_changed:
            checkpoint.pop("optimizer_states", None)


class VocoderGAN(VCVITS):
    """BASELINE.json configs[1]: "HiFi-GAN generator+discriminator step only" -- net_g is the
    decoder alone; the batch feeds `z_slice` [B, inter_channels, 32] and the matching target
    waveform segment `y_wav_values` [B, 1, segment_size] directly."""

    def _build_generator(self):
        m = self.hparams.model
        return Generator(m.inter_channels, m.resblock, m.resblock_kernel_sizes, m.resblock_dilation_sizes,
                         m.upsample_rates, m.upsample_initial_channel, m.upsample_kernel_sizes)

    def _generator_pass(self, batch, decoder_only=False):
        y = batch["y_wav_values"]
        y_hat = self.net_g(batch["z_slice"])
        y_mel = None
        if not decoder_only:
            with torch.no_grad():
                _, y_mel = self._spec_mel(y.squeeze(1))
        return y_hat, y, y_mel, None
