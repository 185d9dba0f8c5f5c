"""TS-ASR Brain: the reference recipes' ``TSASR(sb.Brain)`` on the MI355X runtime.

Mirrors train_librispeechmix_scratch.py:33-195 (and the ``pretrained`` / ``none`` variants, which differ only in
the speaker branch: train_librispeechmix_pretrained.py:45-63, train_librispeechmix_none.py): same module names in
``self.modules``, same call order, same return values (``compute_forward -> (logits [B,T',U+1,V], hyps or None)``,
``compute_objectives -> 0-dim loss with autograd``). Differences: joiner + head run as ONE fused HIP kernel
(the [B,T',U+1,J] joint tensor is never built), lengths stay on the device, plotting/WER bookkeeping is left to the
caller (out of scope).
"""
import os

import torch

from .. import core, prof, rnnt
from ..nnet import abs_lengths_round

Stage = core.Stage


# TSASR_OVERLAP: "1" (default) the speaker branch and, behind it, the predictor run on a second HIP stream; "2" only the speaker branch;
# "0" one stream. (Round 2's other placements - predictor on a third stream, issued after the encoder, its backward run re-entrantly,
# a high-priority side stream - were measured equal or slower and are gone: profiles/r02_notes.md section 5.)
_OVERLAP_MODE = os.environ.get("TSASR_OVERLAP", "1")
_OVERLAP_DEFAULT = _OVERLAP_MODE != "0"
_PRED_STREAM_MIN_U = int(os.environ.get("TSASR_PRED_STREAM_MIN_U", "512"))      # tokens from which the predictor gets a stream of its own


class TSASR(core.Brain):
    """One class for the reference's three recipe scripts; ``variant`` says which speaker branch runs:
    "scratch" (train_librispeechmix_scratch.py), "pretrained" (train_librispeechmix_pretrained.py, frozen speaker encoder) or "none"
    (train_librispeechmix_none.py). Taken from hparams["variant"] / run_opts when given, else from what the YAML declares: the
    pretrained YAML has `speaker_encoder_path` and no speaker front-end (conformer-t_wavlm.yaml:121-123), the `none` YAML no speaker_proj."""

    def __init__(self, modules=None, opt_class=None, hparams=None, run_opts=None, checkpointer=None, profiler=None):
        super().__init__(modules, opt_class, hparams, run_opts, checkpointer, profiler)
        hp, ro = hparams or {}, run_opts or {}
        variant = ro.get("variant", hp.get("variant"))
        if variant is None:
            if "speaker_proj" not in self.modules:
                variant = "none"
            elif "speaker_encoder_path" in hp or "speaker_frontend" not in self.modules:
                variant = "pretrained"
            else:
                variant = "scratch"
        if variant not in ("scratch", "pretrained", "none"):
            raise ValueError(f"unknown TSASR variant {variant!r}")
        self.variant = variant

    # ---- speaker branch (train_librispeechmix_scratch.py:44-80) ------------------------------------
    def _speaker_embedding(self, batch, epoch):
        hp = self.hparams
        if self.variant == "none":
            return None, None
        if self.variant == "pretrained":
            # train_librispeechmix_pretrained.py:45-63,80: the frozen speaker encoder's x-vector [B,1,E] (cross_attention: its last hidden
            # states [B,S,E]) goes through speaker_proj. The embedding arrives with the batch (`enroll_emb`, the synthetic / precomputed
            # form); a module registered as `speaker_encoder` (Hugging Face AutoModelForAudioXVector signature) is run as the reference does.
            if hasattr(batch, "enroll_emb"):
                embs, lens = batch.enroll_emb
            elif "speaker_encoder" in self.modules:
                enroll, lens = batch.enroll_sig
                with torch.no_grad():
                    self.modules.speaker_encoder.eval()
                    L = enroll.shape[-1]
                    n = (lens * L).ceil().clamp(max=L).int()
                    out = self.modules.speaker_encoder(input_values=enroll, attention_mask=(torch.arange(L, device=enroll.device)[None, :] < n[:, None]).long(),
                                                       output_attentions=False, output_hidden_states=hp.injection_mode == "cross_attention")
                embs = (out.hidden_states[-1][..., :hp.speaker_embedding_dim] if hp.injection_mode == "cross_attention"
                        else out.embeddings[:, None, :])
            else:
                raise ValueError("pretrained variant: the batch needs an `enroll_emb` field or modules['speaker_encoder'] must be set")
            return self.modules.speaker_proj(embs), lens
        enroll, enroll_lens = batch.enroll_sig
        if getattr(hp, "input_is_feats", False):
            feats = enroll
        else:
            feats = self.modules.speaker_feature_extractor(enroll)
            feats = self.modules.speaker_normalizer(feats, enroll_lens, epoch=epoch)
        feats = self.modules.speaker_frontend(feats)
        embs = self.modules.speaker_encoder(feats, enroll_lens)
        if hp.injection_mode != "cross_attention":     # masked mean-pool over the first ceil(len * T'e) frames (:52-64): one HIP launch
            from .. import ops
            embs = ops.mean_pool(embs, enroll_lens)
        return self.modules.speaker_proj(embs), enroll_lens

    def _predictor(self, tokens_bos, tokens_bos_lens):
        dec = self.modules.decoder
        if hasattr(dec, "forward_tokens"):       # embedding + decoder in one call: one-hot tokens never become a [B,U,V-1] tensor
            dec_out, _ = dec.forward_tokens(tokens_bos, self.modules.embedding, lengths=tokens_bos_lens)
        else:
            dec_out, _ = dec(self.modules.embedding(tokens_bos), lengths=tokens_bos_lens)
        return self.modules.decoder_proj(dec_out)

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._aux_streams.append(self._side)      # Brain joins it after backward
        return self._side

    def _pred_stream(self):
        if getattr(self, "_third", None) is None:
            self._third = torch.cuda.Stream(device=self.device)
            self._aux_streams.append(self._third)
        return self._third

    def _flush_main_wgrads(self, grad):
        """Tensor hook on the speaker embedding: fires (on the side stream) when the speaker branch's backward is about to start, i.e.
        when the mixture encoder's and front-end's backward have been enqueued on the main stream. Their queued weight gradients
        (three quarters of the step's) are launched there now, as one grouped kernel that runs beside the speaker branch's small,
        latency-bound backward kernels instead of after them; the rest follows in finish_backward. The gradient passes through."""
        prof.stamp("speaker backward starts [side]")
        arena = getattr(self, "arena", None)
        if arena is not None and arena.in_backward and getattr(arena, "_main_stream", None) is not None:
            with torch.cuda.stream(arena._main_stream):
                arena.flush_wgrads(hold=True)
                prof.stamp("early weight gradients done [main]")
        return None

    def compute_forward(self, batch, stage):
        hp = self.hparams
        epoch = hp.epoch_counter.current if hasattr(hp, "epoch_counter") else 0
        batch = batch.to(self.device)
        mixed, mixed_lens = batch.mixed_sig
        tokens_bos, tokens_bos_lens = batch.tokens_bos
        tok = batch.tokens.data      # the loss's targets as the lattice kernels read them (int32): cast here, off the loss's own chain
        self._tokens32 = (tok, tok if tok.dtype == torch.int32 else tok.to(torch.int32).contiguous())
        # Two parts of the forward do not depend on the mixture encoder: the speaker branch (6 encoder layers over the enrollment,
        # half-filled grids; needed at the injection point) and the predictor (embedding -> LSTM -> projection; its persistent
        # recurrence kernels occupy 32 of 256 CUs for 0.4 + 0.7 ms; needed at the joint). Both run on a second HIP stream; the
        # encoder joins the first lazily at the injection, the joint waits for the second. Backward follows by itself (autograd
        # runs a node on the stream of its forward): the predictor's backward overlaps the last encoder layers', the speaker
        # branch's overlaps layer 0 / the front-end's. Works the same inside a captured hipGraph (fork / join become edges).
        overlap = (stage == Stage.TRAIN and self.variant != "none" and getattr(self, "overlap_branches", _OVERLAP_DEFAULT)
                   and torch.device(self.device).type == "cuda")
        dec_out = None
        if overlap:
            cur, side = torch.cuda.current_stream(), self._side_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                spk, enroll_lens = self._speaker_embedding(batch, epoch)
                spk_ready = torch.cuda.Event()
                spk_ready.record(side)
                prof.stamp("speaker forward done [side]")
                if spk is not None and spk.requires_grad:
                    spk.register_hook(self._flush_main_wgrads)
                # The predictor follows the speaker branch on the forked stream, in eager and in captured steps alike (one autograd
                # topology: gradient buckets complete in the same order either way). Round 2 kept it on the main stream in eager steps
                # because runs with it forked deviated once in ~100; the cause was found in round 3 and is not a stream-ordering
                # matter: packed-fp32 instructions of one kernel returning wrong values while the per-step LSTM kernels ran beside it
                # (profiles/r03_notes.md section 1; the library no longer contains such instructions). "2" = speaker branch only.
                # Long targets (configs[4]: U = 1920, a 4.4 ms walk of 1920 dependent steps at B = 1): behind the speaker branch the
                # predictor ends 1.1 ms after the mixture encoder and the joint waits for it; on a stream of its own it starts with the
                # step. Short targets keep the one side stream: at configs[1] a third stream cost 0.1 ms (profiles/r04_notes.md section 4).
                own = _OVERLAP_MODE != "2" and tokens_bos.shape[1] >= _PRED_STREAM_MIN_U
                if _OVERLAP_MODE != "2" and not own:
                    dec_out = self._predictor(tokens_bos, tokens_bos_lens)
                    prof.stamp("predictor forward done [side]")
                    if prof.STAMPS and dec_out.requires_grad:
                        dec_out.register_hook(lambda g: prof.stamp("predictor backward starts [side]"))
            if own:
                third = self._pred_stream()
                third.wait_stream(cur)
                with torch.cuda.stream(third):
                    dec_out = self._predictor(tokens_bos, tokens_bos_lens)
                    prof.stamp("predictor forward done [third]")
                    if prof.STAMPS and dec_out.requires_grad:
                        dec_out.register_hook(lambda g: prof.stamp("predictor backward starts [third]"))

            def speaker_embs():
                prof.stamp("mixture reaches the injection [main]")
                cur.wait_event(spk_ready)
                spk.record_stream(cur)
                return spk
        else:
            speaker_embs, enroll_lens = self._speaker_embedding(batch, epoch)

        augment = bool(getattr(hp, "augment", False)) and stage == Stage.TRAIN
        if getattr(hp, "input_is_feats", False):
            feats = mixed
        else:
            if augment and "speed_perturb" in self.modules:      # train_librispeechmix_scratch.py:82-85
                mixed = self.modules.speed_perturb(mixed)
            feats = self.modules.feature_extractor(mixed)
            feats = self.modules.normalizer(feats, mixed_lens, epoch=epoch)
        if augment and "augmentation" in self.modules:           # train_librispeechmix_scratch.py:91-94
            feats = self.modules.augmentation(feats)
        feats = self.modules.frontend(feats)
        prof.stamp("mixture front-end forward done [main]")
        if prof.STAMPS and feats.requires_grad:
            feats.register_hook(lambda g: prof.stamp("mixture encoder backward done [main]"))
        enc_out = self.modules.encoder(feats, mixed_lens, speaker_embs, enroll_lens)
        enc_out = self.modules.encoder_proj(enc_out)
        prof.stamp("mixture encoder forward done [main]")
        if prof.STAMPS and enc_out.requires_grad:
            enc_out.register_hook(lambda g: prof.stamp("joint backward done [main]"))

        if dec_out is None:
            dec_out = self._predictor(tokens_bos, tokens_bos_lens)
        else:
            cur.wait_stream(side)
            if own:
                cur.wait_stream(third)
            dec_out.record_stream(cur)

        # joiner + transducer_head fused (train_librispeechmix_scratch.py:132-135)
        head = self.modules.transducer_head.w
        tlen = abs_lengths_round(mixed_lens, enc_out.shape[1])
        ulen = abs_lengths_round(batch.tokens.lengths.to(self.device), batch.tokens.data.shape[1])
        logits = rnnt.fused_joint_logits(enc_out, dec_out, head.weight, head.bias, self.modules.joiner.nonlinearity.negative_slope,
                                         tlen, ulen)
        prof.stamp("joint forward done [main]")
        hyps = None
        if stage == Stage.VALID:
            if epoch % getattr(hp, "valid_search_freq", 1) == 0 and hasattr(hp, "greedy_searcher"):
                hyps, _, _, _ = hp.greedy_searcher(enc_out)
        elif stage == Stage.TEST and hasattr(hp, "beam_searcher"):
            hyps, _, _, _ = hp.beam_searcher(enc_out)
        return logits, hyps

    def compute_objectives(self, predictions, batch, stage):
        logits, hyps = predictions
        _, mixed_lens = batch.mixed_sig
        tokens, tokens_lens = batch.tokens
        t32 = getattr(self, "_tokens32", None)
        if t32 is not None and t32[0].data_ptr() == tokens.data_ptr() and t32[0].shape == tokens.shape:
            tokens = t32[1]
        loss = self.hparams.transducer_loss(logits, tokens, mixed_lens, tokens_lens)
        prof.stamp("loss forward done [main]")
        if hyps is not None:
            self.last_hyps = hyps  # the reference feeds them to its WER/CER statistics (out of scope here)
        return loss

    def on_fit_batch_end(self, batch, outputs, loss, should_step):
        if getattr(self.hparams, "enable_scheduler", False) and should_step:
            self.hparams.noam_scheduler(self.optimizer)
