"""speechbrain.inference.ASR mirror: EncoderDecoderASR (inference/ASR.py:35-173)."""
from dataclasses import dataclass as _dataclass
from typing import Optional as _Optional

import torch

from speechbrain_amd.inference.interfaces import Pretrained


class EncoderDecoderASR(Pretrained):
    """Same surface as the reference: transcribe_file / transcribe_batch / encode_batch / forward.

    ``modules`` needs ``encoder`` (Fbank -> InputNormalization -> ConvolutionFrontEnd container) and
    ``decoder`` (a searcher); with ``hparams['transformer_beam_search']`` true, ``modules['transformer']``
    (TransformerASR) encodes the front-end output (ASR.py:70-74,126-128)."""

    HPARAMS_NEEDED = ["tokenizer"]
    MODULES_NEEDED = ["encoder", "decoder"]

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.tokenizer = self.hparams.tokenizer
        self.transducer_beam_search = getattr(self.hparams, "transducer_beam_search", False)
        self.transformer_beam_search = getattr(self.hparams, "transformer_beam_search", False)
        if self.transducer_beam_search:
            raise NotImplementedError("transducer decoding is a different model family")

    def transcribe_file(self, path, **kwargs):
        waveform = self.load_audio(path, **kwargs)
        words, _ = self.transcribe_batch(waveform.unsqueeze(0), torch.tensor([1.0]))
        return words[0]

    def encode_batch(self, wavs, wav_lens):
        from speechbrain_amd import native

        wavs = wavs.float()
        wavs, wav_lens = wavs.to(self.device), wav_lens.to(self.device)
        encoder_out = self.mods.encoder(wavs, wav_lens)  # Fbank / normalisation / CNN: always fp32
        if self.transformer_beam_search:
            with native.precision_scope(self.eval_precision):
                encoder_out = self.mods.transformer.encode(encoder_out, wav_lens)
        return encoder_out

    def encode_group(self, batches):
        """``encode_batch`` of several independently padded batches [(wavs, wav_lens), ...] -> [encoder_out, ...].
        The feature front-end runs batch by batch; the Conformer encoder runs once over the rows of all of them
        (TransformerASR.encode_group), so its GEMMs are large enough to fill the chip even with recipe-sized
        batches.  Results equal ``encode_batch`` batch by batch."""
        from speechbrain_amd import native

        if not (self.transformer_beam_search and hasattr(self.mods.transformer, "encode_group")):
            return [self.encode_batch(w, l) for w, l in batches]
        feats, lens = [], []
        for wavs, wav_lens in batches:
            wavs, wav_lens = wavs.float().to(self.device), wav_lens.to(self.device)
            feats.append(self.mods.encoder(wavs, wav_lens))
            lens.append(wav_lens)
        with native.precision_scope(self.eval_precision):
            return self.mods.transformer.encode_group(feats, lens)

    def transcribe_batch(self, wavs, wav_lens):
        with torch.no_grad():
            wav_lens = wav_lens.to(self.device)
            encoder_out = self.encode_batch(wavs, wav_lens)
            predicted_tokens, _, _, _ = self.mods.decoder(encoder_out, wav_lens)
            predicted_words = [self.tokenizer.decode_ids(seq) for seq in predicted_tokens]
        return predicted_words, predicted_tokens

    def forward(self, wavs, wav_lens):
        return self.transcribe_batch(wavs, wav_lens)


@_dataclass
class ASRWhisperSegment:
    """inference/ASR.py:391-428: one chunk of a long-form Whisper transcription."""

    start: float
    end: float
    chunk: torch.Tensor
    lang_id: _Optional[str] = None
    words: _Optional[str] = None
    tokens: _Optional[list] = None
    prompt: _Optional[list] = None
    avg_log_probs: _Optional[float] = None
    no_speech_prob: _Optional[float] = None


class WhisperASR(Pretrained):
    """inference/ASR.py:431-945, the batch entry points: ``mods.whisper`` (integrations.huggingface.whisper.Whisper)
    and ``mods.decoder`` (a Whisper searcher, e.g. S2SWhisperGreedySearcher).  ``encode_batch`` = log-mel (padded or
    trimmed to the 30-second chunk) -> Whisper encoder; ``transcribe_batch`` adds the search and the tokenizer's
    decoding; ``detect_language_*``, ``transcribe_file`` / ``transcribe_file_streaming`` (30-second segments, running
    prompt, no-speech rule) as in the reference, reading the file with the built-in wav reader."""

    HPARAMS_NEEDED = ["language", "sample_rate"]
    MODULES_NEEDED = ["whisper", "decoder"]
    TASKS = ["transcribe", "translate", "lang_id"]

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.tokenizer = self.hparams.whisper.tokenizer

    # ---- the encoder under the interface's precision (inference/interfaces.py:295-298: every forward of a Pretrained
    # runs inside the context built from run_opts; here that context is native.precision_scope) -----------------------
    def _mel(self, wavs):
        return self.mods.whisper._get_mel(wavs.to(device=self.device, dtype=torch.float32))

    def _encode_mel(self, mel):
        from speechbrain_amd import native

        with native.precision_scope(self.eval_precision):
            return self.mods.whisper.forward_encoder(mel)

    def encode_batch(self, wavs, wav_lens):
        return self._encode_mel(self._mel(wavs))

    @torch.no_grad()
    def transcribe_batch(self, wavs, wav_lens):
        wav_lens = wav_lens.float().to(self.device)
        predicted_tokens, _, _, _ = self.mods.decoder(self.encode_batch(wavs, wav_lens), wav_lens)
        predicted_words = [self._text(t) for t in predicted_tokens]
        if getattr(self.hparams, "normalized_transcripts", False):
            predicted_words = [self.tokenizer.normalize(text).split(" ") for text in predicted_words]
        return predicted_words, predicted_tokens

    def forward(self, wavs, wav_lens):
        return self.transcribe_batch(wavs, wav_lens)

    def _text(self, tokens):
        return self.tokenizer.decode(tokens, skip_special_tokens=True).strip()

    # ---- language identification (inference/ASR.py:475-560) --------------------------------------------------------
    @torch.no_grad()
    def detect_language_file(self, path: str):
        """Language of an audio file of 30 seconds or less: (language tokens [1], [{code: probability}])."""
        return self.mods.whisper.detect_language(self._mel(self.load_audio(path).unsqueeze(0)))

    @torch.no_grad()
    def detect_language_batch(self, wav):
        return self.mods.whisper.detect_language(self._mel(wav))

    def _segment_languages(self, mel, task):
        """The language code of every item of ``mel``: the model's configured language, unless none is configured or the
        task is language identification -- then the most probable detected one, whose tokens also become the
        searcher's language tokens (inference/ASR.py:541-560)."""
        whisper = self.mods.whisper
        if whisper.language is not None and task != "lang_id":
            return [whisper.language] * mel.shape[0]
        lang_tokens, lang_probs = whisper.detect_language(mel)
        self.mods.decoder.set_lang_tokens(lang_tokens)
        return [max(probs, key=probs.get) for probs in lang_probs]

    # ---- long-form transcription (inference/ASR.py:622-840) ------------------------------------------------------
    def _file_segments(self, path, chunk_size, **kwargs):
        """(start_s, end_s, samples [1, n]) for consecutive ``chunk_size``-second pieces of the file."""
        audio = self.load_audio(path, **kwargs).unsqueeze(0)
        pieces = split_fixed_chunks(audio, chunk_size * self.hparams.sample_rate)
        for k, piece in enumerate(pieces):
            yield k * chunk_size, (k + 1) * chunk_size, piece.to(self.device)

    @torch.no_grad()
    def transcribe_file_streaming(self, path, task=None, initial_prompt=None, logprob_threshold=-1.0,
                                  no_speech_threshold=0.6, condition_on_previous_text=False, verbose=False,
                                  use_torchaudio_streaming=False, chunk_size=30, **kwargs):
        """Yields one ``ASRWhisperSegment`` per ``chunk_size``-second segment of the file (tasks ``transcribe``,
        ``translate``, ``lang_id``): log-mel -> encoder -> language -> search with the running prompt -> the no-speech /
        average-log-prob skip rule.  The file is read whole with the built-in wav reader and cut into segments (the
        reference's ffmpeg-backed ``torchaudio.io.StreamReader`` path, ``use_torchaudio_streaming=True``, is not part
        of this package; the segments and therefore the results are the same)."""
        if use_torchaudio_streaming:
            raise NotImplementedError("torchaudio.io.StreamReader (ffmpeg) streaming: pass use_torchaudio_streaming=False")
        if task is not None and task not in self.TASKS:
            raise ValueError(f"Task {task} not supported. Supported tasks are {self.TASKS}")
        searcher = self.mods.decoder
        if task in ("transcribe", "translate"):
            searcher.set_task(task)
        history = _PromptHistory(self.tokenizer, initial_prompt,
                                 carry_over=lambda: condition_on_previous_text and searcher.temperature <= 0.5)
        whole = torch.tensor([1.0])
        for start, end, samples in self._file_segments(path, chunk_size, **kwargs):
            mel = self._mel(samples)
            memory = self._encode_mel(mel)
            lang = self._segment_languages(mel, task)[0]
            seg = ASRWhisperSegment(start=start, end=end, chunk=samples, lang_id=lang)
            if task == "lang_id":
                yield seg
                continue
            seg.prompt = history.current()
            searcher.set_prompt(seg.prompt)
            hyps, _, scores, _ = searcher(memory, whole)
            seg.avg_log_probs = float(scores.sum() / (len(hyps[0]) + 1))
            seg.no_speech_prob = searcher.no_speech_probs[0]
            if _is_silence(seg.no_speech_prob, seg.avg_log_probs, no_speech_threshold, logprob_threshold):
                seg.words, seg.tokens = "", []
                yield seg
                continue
            seg.words, seg.tokens = self._text(hyps[0]), hyps[0]
            yield seg
            history.append(hyps[0])

    def transcribe_file(self, path, task=None, initial_prompt=None, logprob_threshold=-1.0, no_speech_threshold=0.6,
                        condition_on_previous_text=False, verbose=False, use_torchaudio_streaming=False, chunk_size=30,
                        **kwargs):
        """The list of ``ASRWhisperSegment`` of the whole file (:790-865)."""
        results = []
        for seg in self.transcribe_file_streaming(
                path, task=task, initial_prompt=initial_prompt, logprob_threshold=logprob_threshold,
                no_speech_threshold=no_speech_threshold, condition_on_previous_text=condition_on_previous_text,
                verbose=verbose, use_torchaudio_streaming=use_torchaudio_streaming, chunk_size=chunk_size, **kwargs):
            results.append(seg)
            if verbose:
                print(f"[{seg.start}s --> {seg.end}s] {seg.words if task != 'lang_id' else seg.lang_id}")
        return results


def _is_silence(no_speech_prob, avg_log_prob, no_speech_threshold, logprob_threshold):
    """The segment skip rule of long-form Whisper decoding (inference/ASR.py:742-757): a segment is dropped when the
    no-speech probability exceeds its threshold -- unless the hypothesis' average log-probability clears its own."""
    if no_speech_threshold is None or not no_speech_prob > no_speech_threshold:
        return False
    return logprob_threshold is None or not avg_log_prob > logprob_threshold


class _PromptHistory:
    """The token history that conditions the next segment's search (inference/ASR.py:690-700, 775-787): everything decoded
    so far (after an optional initial prompt), of which only the part behind ``mark`` is shown to the searcher.  Unless
    ``carry_over()`` holds after a segment, the mark moves to the end, i.e. the next segment starts from an empty prompt."""

    def __init__(self, tokenizer, initial_prompt, carry_over):
        self.tokens = list(tokenizer.encode(" " + initial_prompt.strip())) if initial_prompt is not None else []
        self.mark, self.carry_over = 0, carry_over

    def current(self):
        return self.tokens[self.mark:]

    def append(self, hyp):
        self.tokens.extend(hyp)
        if not self.carry_over():
            self.mark = len(self.tokens)


# ---------------------------------------------------------------------------------------------- streaming
from dataclasses import dataclass  # noqa: E402
from itertools import chain  # noqa: E402
from typing import Any, List, Optional  # noqa: E402


def split_fixed_chunks(x, chunk_size, dim=-1):
    """utils/streaming.py split_fixed_chunks: consecutive chunks of `chunk_size` along `dim` (the last may be short)."""
    n = x.shape[dim]
    return [x.narrow(dim, t0, min(chunk_size, n - t0)) for t0 in range(0, n, chunk_size)]


@dataclass
class ASRStreamingContext:
    """inference/ASR.py:948-975: the mutable state of one streaming session."""

    config: Any                              # DynChunkTrainConfig; fixed for the session
    fea_extractor_context: Any
    encoder_context: Any
    decoder_context: Any
    tokenizer_context: Optional[List[Any]]   # one per batch item, created at the first decode


class StreamingASR(Pretrained):
    """Chunk-by-chunk ASR (inference/ASR.py:978-1363): ``hparams.fea_streaming_extractor`` (a
    StreamingFeatureWrapper) -> ``mods.enc.forward_streaming`` (Conformer with left-context caches) ->
    ``mods.proj_enc`` -> ``hparams.decoding_function``.  The decoder of the reference's streaming models is a
    transducer search, which is a different model family; any callable ``decoding_function(x, decoder_context) ->
    list[list[int]]`` plugs in (e.g. a greedy CTC decoder over ``proj_enc``'s output)."""

    HPARAMS_NEEDED = ["fea_streaming_extractor", "make_decoder_streaming_context", "decoding_function",
                      "make_tokenizer_streaming_context", "tokenizer_decode_streaming"]
    MODULES_NEEDED = ["enc", "proj_enc"]

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.filter_props = self.hparams.fea_streaming_extractor.properties

    def make_streaming_context(self, dynchunktrain_config):
        return ASRStreamingContext(
            config=dynchunktrain_config,
            fea_extractor_context=self.hparams.fea_streaming_extractor.make_streaming_context(),
            encoder_context=self.mods.enc.make_streaming_context(dynchunktrain_config),
            decoder_context=self.hparams.make_decoder_streaming_context(), tokenizer_context=None)

    def get_chunk_size_frames(self, dynchunktrain_config) -> int:
        """Input samples per chunk, as the reference computes it (:1236-1252)."""
        return (self.filter_props.stride - 1) * dynchunktrain_config.chunk_size

    @torch.no_grad()
    def encode_chunk(self, context: ASRStreamingContext, chunk, chunk_len=None):
        if chunk_len is None:
            chunk_len = torch.ones((chunk.size(0),))
        chunk, chunk_len = chunk.float().to(self.device), chunk_len.to(self.device)
        assert chunk.shape[-1] <= self.get_chunk_size_frames(context.config)
        x = self.hparams.fea_streaming_extractor(chunk, context=context.fea_extractor_context, lengths=chunk_len)
        x = self.mods.enc.forward_streaming(x, context.encoder_context)
        return self.mods.proj_enc(x)

    @torch.no_grad()
    def decode_chunk(self, context: ASRStreamingContext, x):
        tokens = self.hparams.decoding_function(x, context.decoder_context)
        if context.tokenizer_context is None:
            context.tokenizer_context = [self.hparams.make_tokenizer_streaming_context() for _ in range(len(tokens))]
        words = [self.hparams.tokenizer_decode_streaming(self.hparams.tokenizer, cur, context.tokenizer_context[i])
                 for i, cur in enumerate(tokens)]
        return words, tokens

    def transcribe_chunk(self, context: ASRStreamingContext, chunk, chunk_len=None):
        words, _ = self.decode_chunk(context, self.encode_chunk(context, chunk, chunk_len))
        return words

    def transcribe_file_streaming(self, path, dynchunktrain_config, use_torchaudio_streaming: bool = True, **kwargs):
        """Yields the text of every chunk.  The file is read with the built-in wav reader and cut into chunks
        (the reference's ffmpeg-backed ``torchaudio.io.StreamReader`` is not part of this package; the chunks and
        therefore the results are the same)."""
        chunk_size = self.get_chunk_size_frames(dynchunktrain_config)
        chunks = split_fixed_chunks(self.load_audio(path, **kwargs).unsqueeze(0), chunk_size)
        rel_length = torch.tensor([1.0])
        context = self.make_streaming_context(dynchunktrain_config)
        n_final = self.hparams.fea_streaming_extractor.get_recommended_final_chunk_count(chunk_size)
        final_chunks = [torch.zeros((1, chunk_size), device=self.device)] * n_final
        for chunk in chain(chunks, final_chunks):
            yield self.transcribe_chunk(context, chunk, rel_length)[0]

    def transcribe_file(self, path, dynchunktrain_config, use_torchaudio_streaming: bool = True):
        return "".join(self.transcribe_file_streaming(path, dynchunktrain_config, use_torchaudio_streaming))
