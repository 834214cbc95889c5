"""speechbrain.decoders.seq2seq mirror: S2STransformerBeamSearcher / S2STransformerGreedySearcher.

Constructor signatures, ``forward(enc_states, wav_len)`` and return tuples follow the reference
(decoders/seq2seq.py:711-1934).  One forward = one C-ABI call that runs the whole search on the
device (csrc/search.hip): KV-cached decoder steps, fused CTC prefix scoring, device-side beam
bookkeeping; the only device->host traffic is the stop-rule poll and the final token ids.
"""
import ctypes

import torch

from speechbrain_amd import native


class S2SBaseSearcher(torch.nn.Module):
    def __init__(self, bos_index, eos_index, min_decode_ratio, max_decode_ratio):
        super().__init__()
        self.bos_index, self.eos_index = bos_index, eos_index
        self.min_decode_ratio, self.max_decode_ratio = min_decode_ratio, max_decode_ratio

    def _handle(self):
        """Weight table for the C ABI; rebuilt when any source parameter moved OR was updated in place
        (the table holds LayerNorm-folded copies of the projections)."""
        h = getattr(self, "_dec_handle", None)
        if h is None or h.stale(self.model, self.fc):
            h = native.DecoderHandle(self.model, self.fc)
            self._dec_handle = h
        return h

    def _steps(self, T):
        # seq2seq.py:1336-1338 / :218-219
        return int(T * self.min_decode_ratio), int(T * self.max_decode_ratio)


class S2STransformerGreedySearcher(S2SBaseSearcher):
    """seq2seq.py:330-367 on top of S2SGreedySearcher.forward (:176-327), temperature 0 (arg-max).

    Returns (hyps, top_lengths [B,1], top_scores [B,1,L], None): the reference's fourth output, the
    full [B,1,L,V] log-probability tensor, is not materialised on the device path."""

    def __init__(self, modules, temperature=0.0, **kwargs):
        super().__init__(**kwargs)
        if temperature != 0.0:
            raise NotImplementedError("sampling (temperature > 0) is not on the ASR inference path")
        self.model, self.fc, self.temperature = modules[0], modules[1], temperature
        self.check_every = 8

    @torch.no_grad()
    def forward(self, enc_states, wav_len, attention_mask=None):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is a Whisper/LLM decoder feature")
        B, T, _ = enc_states.shape
        enc_lens = torch.round(T * wav_len).int().to(enc_states.device)
        mn, mx = self._steps(T)
        tok, sc, steps = native.greedy_search(self._handle(), enc_states.contiguous(), enc_lens, mn, mx, self.bos_index,
                                              self.eos_index, self.check_every)
        tok, sc = tok[:, :steps].cpu(), sc[:, :steps].cpu()
        # the reference stops at the first step after which every utterance has ended (:277-278)
        is_eos = tok == self.eos_index
        first = torch.where(is_eos.any(1), is_eos.float().argmax(1), torch.full((B,), steps))
        L = steps if (first == steps).any() else int(first.max()) + 1
        L = max(L, 1) if steps > 0 else 0
        tok, sc = tok[:, :L], sc[:, :L]
        hyps, rel = [], []
        for b in range(B):
            n = min(int(first[b]), L)
            rel.append(n / L if L else 0.0)
            hyps.append(tok[b, :n].tolist())
        return hyps, torch.tensor(rel).unsqueeze(1), sc.unsqueeze(1), None


class _WhisperPrompting:
    """What S2SWhisperGreedySearcher (seq2seq.py:421-636) and S2SWhisperBeamSearcher (:1937-2206) share: the initial
    tokens (prefix / prompt / task / language), the suppression masks and the setters."""

    def _refresh(self):
        self.initial_tokens = self._get_initial_tokens()
        self.sample_begin = len(self.initial_tokens)
        self.eos_index, self.bos_index = self.model.eos, self.initial_tokens[-1]

    def set_lang_tokens(self, lang_tokens):
        self.lang_tokens = lang_tokens

    def set_task(self, task):
        self.model.set_task(task)
        self._refresh()

    def set_prompt(self, prompt):
        self.prompt = prompt
        self._refresh()

    @property
    def get_tokens_to_suppress(self):
        """:512-540: the configured list ("-1" = the model's non-speech symbols) plus the task / start tokens."""
        sup = self.suppress_tokens
        if isinstance(sup, str):
            sup = [int(t) for t in sup.split(",")]
        if sup is None or len(sup) == 0:
            sup = []
        elif -1 in sup:
            sup = [t for t in sup if t >= 0] + list(self.model.non_speech_tokens)
        sup = list(sup) + [self.model.transcribe, self.model.translate, self.model.bos, self.model.bos_prev, self.model.bos_lm]
        return tuple(sorted(set(sup)))

    def _get_initial_tokens(self):
        """:542-573."""
        tok = self.model.tokenizer
        tokens = list(tok.prefix_tokens)
        if self.prefix:
            pre = tok.encode(" " + self.prefix.strip(), add_special_tokens=False) if isinstance(self.prefix, str) else list(self.prefix)
            if self.sample_len is not None:
                pre = pre[-(self.max_attn_tokens // 2 - self.sample_len):]
            tokens = tokens + pre
        if self.prompt:
            pro = tok.encode(" " + self.prompt.strip(), add_special_tokens=False) if isinstance(self.prompt, str) else list(self.prompt)
            tokens = [self.model.bos_prev] + pro[-(self.max_attn_tokens // 2 - 1):] + tokens
        return tuple(tokens)

    def _masks(self, V, device):
        key = (self.get_tokens_to_suppress if self.suppress_tokens else (), self.suppress_blank, self.eos_index, V, str(device))
        if getattr(self, "_mask_key", None) != key:
            always = torch.zeros(V)
            if self.suppress_tokens:
                always[list(self.get_tokens_to_suppress)] = -float("inf")
            first = None
            if self.suppress_blank:  # :619-625: no blank and no EOS as the very first sampled token
                first = torch.zeros(V)
                first[list(self.model.tokenizer.encode(" ", add_special_tokens=False)) + [self.eos_index]] = -float("inf")
                first = first.to(device)
            self._mask_key, self._mask = key, (always.to(device) if self.suppress_tokens else None, first)
        return self._mask


class S2SWhisperGreedySearcher(_WhisperPrompting, S2SBaseSearcher):
    """seq2seq.py:421-636 (on S2SGreedySearcher.forward, :176-327): greedy decoding of a
    ``speechbrain_amd.integrations.huggingface.whisper.Whisper`` -- initial tokens (prefix / prompt / task /
    language) prime the device-side KV cache, then arg-max steps with the blank / non-speech / special-token masks,
    all inside one C-ABI call (sbk_prompted_greedy_search_f32).

    forward(enc_states, wav_len) -> (hyps, top_lengths [B,1], top_scores [B,1,L], None); ``no_speech_probs`` is set as
    in the reference.  The fourth output of the reference, the full [B,1,L,V] log-probability tensor, is not
    materialised on the device path."""

    def __init__(self, model, temperature=0.0, use_kv_cache=True, suppress_blank=True, suppress_tokens="-1",
                 sample_len=None, prefix=None, prompt=None, **kwargs):
        kwargs.setdefault("min_decode_ratio", 0.0)
        kwargs.setdefault("max_decode_ratio", 1.0)
        super().__init__(bos_index=model.bos, eos_index=model.eos, **kwargs)
        if temperature != 0.0:
            raise NotImplementedError("sampling (temperature > 0) is not on the ASR inference path")
        self.model, self.temperature, self.use_kv_cache = model, temperature, use_kv_cache
        self.suppress_blank, self.suppress_tokens = suppress_blank, suppress_tokens
        self.prefix, self.prompt = prefix, prompt
        cfg = model.model.config
        self.max_attn_tokens = cfg.get("max_length", cfg.get("max_target_positions", 448))
        self.sample_len = sample_len or self.max_attn_tokens // 2
        self.no_speech_probs, self.lang_tokens = None, None
        self.check_every = 8
        self._refresh()

    @torch.no_grad()
    def forward(self, enc_states, wav_len=None, attention_mask=None):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is an LLM decoder feature")
        enc = enc_states.float().contiguous()
        B, T, _ = enc.shape
        dev = enc.device
        P = self.sample_begin
        prompt = torch.tensor([list(self.initial_tokens)] * B, dtype=torch.int32)
        if self.lang_tokens is not None:  # :586-591: per-utterance language token right after <|startoftranscript|>
            prompt[:, self.initial_tokens.index(self.model.bos) + 1] = torch.as_tensor(self.lang_tokens).to(torch.int32).cpu()
        mn, mx = self._steps(T)
        # the loop of the reference (:230-279) runs steps mn..mx-1 and stops once the token memory holds
        # max_attn_tokens - sample_begin entries (:633-635); the memory starts with sample_begin - 1 of them
        max_new = min(mx - mn, max(1, self.max_attn_tokens - 2 * P + 1))  # (one step runs before the end condition fires)
        if max_new <= 0:
            raise ValueError("min_decode_ratio / max_decode_ratio leave no decoding step")
        handle = self.model.decoder_handle()
        always, first = self._masks(handle.W.vocab, dev)
        full = torch.full((B,), T, dtype=torch.int32, device=dev)
        tok, sc, steps, probe = native.prompted_greedy_search(
            handle, enc, full, prompt.to(dev), max_new, self.eos_index, always, first,
            probe=(self.initial_tokens.index(self.model.bos), self.model.no_speech), check_every=self.check_every)
        self.no_speech_probs = probe.cpu().tolist()
        tok, sc = tok[:, :steps].cpu(), sc[:, :steps].cpu()
        is_eos = tok == self.eos_index
        first_eos = torch.where(is_eos.any(1), is_eos.float().argmax(1), torch.full((B,), steps))
        L = steps if (first_eos == steps).any() else int(first_eos.max()) + 1  # stops once every utterance has ended
        tok, sc = tok[:, :L], sc[:, :L]
        hyps, rel = [], []
        for b in range(B):
            n = min(int(first_eos[b]), L)
            rel.append(n / L if L else 0.0)
            hyps.append(tok[b, :n].tolist())
        return hyps, torch.tensor(rel).unsqueeze(1), sc.unsqueeze(1), None


class S2STransformerBeamSearcher(S2SBaseSearcher):
    """seq2seq.py:1853-1934 + S2SBeamSearcher (:711-1749).

    forward(enc_states [B,T',d], wav_len [B]) -> (hyps list[list[int]], best_lens [B], best_scores [B],
    best_log_probs [B,Lmax])."""

    def __init__(self, modules, temperature=1.0, bos_index=None, eos_index=None, min_decode_ratio=0.0,
                 max_decode_ratio=1.0, beam_size=None, scorer=None, return_topk=False, topk=1,
                 using_eos_threshold=True, eos_threshold=1.5, length_normalization=True, using_max_attn_shift=False,
                 max_attn_shift=60, minus_inf=-1e20):
        super().__init__(bos_index, eos_index, min_decode_ratio, max_decode_ratio)
        if topk < 1 or (beam_size is not None and topk > beam_size):
            raise ValueError("topk must lie in [1, beam_size]")
        if using_max_attn_shift:
            raise NotImplementedError("max_attn_shift applies to RNN attention decoders")
        self.model, self.fc, self.temperature = modules[0], modules[1], temperature
        self.beam_size, self.scorer = beam_size, scorer
        self.return_topk, self.topk = return_topk, topk
        self.length_normalization = length_normalization
        self.using_eos_threshold, self.eos_threshold = using_eos_threshold, eos_threshold
        self.minus_inf = minus_inf
        self.attn_weight, self.ctc_weight = 1.0, 0.0
        self.check_every = 8
        self.overlap_ctc = 3  # CTC scorer on a helper stream beside the decoder step (bit mask, see include/sbk.h)
        self.graph_mode = 0   # 1: replay two decoding steps from a captured hipGraph (needs overlap_ctc = 0)
        self.blank_index = 0
        self.ctc_fc = None
        self.ctc_candidates = 0  # > 0: CTC is a partial scorer over that many candidates per hypothesis
        self.ctc_window_size = 0
        self.lm, self.lm_weight, self.lm_temperature = None, 0.0, 1.0
        if scorer is not None:
            if scorer.weights["transformerlm"] != 0.0 and "transformerlm" in scorer.full_scorers:
                lms = scorer.full_scorers["transformerlm"]
                self.lm, self.lm_weight, self.lm_temperature = lms.lm, scorer.weights["transformerlm"], lms.temperature
            if length_normalization and scorer.weights["length"] > 0.0:
                raise ValueError("Length normalization is not compatible with length rewarding.")
            if scorer.weights["ctc"] > 0.0:
                ctc = {**scorer.full_scorers, **scorer.partial_scorers}["ctc"]
                if len({bos_index, eos_index, ctc.blank_index}) < 3:
                    raise ValueError(
                        "Set blank, eos and bos to different indexes for joint ATT/CTC or CTC decoding")
                self.ctc_weight = scorer.weights["ctc"]
                self.attn_weight = 1.0 - self.ctc_weight
                self.blank_index, self.ctc_fc = ctc.blank_index, ctc.ctc_fc
                self.ctc_window_size = int(getattr(ctc, "ctc_window_size", 0))
                if "ctc" in scorer.partial_scorers:  # scorer.py:1287-1291
                    self.ctc_candidates = max(1, int(beam_size * scorer.scorer_beam_scale))
                    if self.ctc_candidates < beam_size:
                        # with fewer candidates than beams a NON-candidate token can be selected; the reference then
                        # advances it with candidate 0's CTC state (score_index -1 -> 0, ctc.py:262-295), the device
                        # search with its own true state: results would differ
                        raise NotImplementedError("CTC as a partial scorer needs scorer_beam_scale >= 1 "
                                                  "(int(beam_size * scorer_beam_scale) candidates >= beam_size)")
        if self.attn_weight <= 0:
            # the reference skips forward_step when attn_weight = 0 and then carries the PREVIOUS step's combined score
            # matrix -- rows still in the previous step's hypothesis order -- into the next step as its "log_probs"
            # (seq2seq.py:916-921 with :1540-1545, :1590-1592): scores accumulate over steps, unpermuted.  Checked against
            # the reference itself (round 4): a plain "CTC scores only" search gives other hypotheses.  Not reproduced.
            raise NotImplementedError("pure-CTC beam search (ctc_weight = 1): the reference's behaviour there is an "
                                      "accumulation quirk of its step loop, not reproduced")

    def config(self, T):
        mn, mx = self._steps(T)
        lm = {}
        if self.lm is not None:
            self._lm_handle = self.lm.handle()  # keeps the pointed-to weight table alive during the call
            self._lm_handle.ready.wait(self._lm_handle.device)  # (built on another worker's stream a moment ago?)
            lm = dict(lm=ctypes.pointer(self._lm_handle.W), lm_weight=self.lm_weight,
                      lm_temperature=self.lm_temperature)
        return native.SearchConfig(**lm, topk=self.topk if self.return_topk else 1,
                                   graph_mode=0 if self.ctc_window_size else int(self.graph_mode),
                                   bos=self.bos_index, eos=self.eos_index, blank=self.blank_index, beam=self.beam_size,
                                   min_steps=mn, max_steps=mx, length_normalization=int(self.length_normalization),
                                   using_eos_threshold=int(self.using_eos_threshold), check_every=self.check_every,
                                   overlap_ctc=0 if self.ctc_window_size else int(self.overlap_ctc),
                                   ctc_candidates=int(self.ctc_candidates), ctc_window_size=int(self.ctc_window_size),
                                   ctc_weight=self.ctc_weight, temperature=self.temperature,
                                   eos_threshold=self.eos_threshold, minus_inf=self.minus_inf)

    @torch.no_grad()
    def search_device(self, enc_states, wav_len):
        """Device-resident results: (tokens [B,L] int32, lens [B] int32, scores [B], log_probs [B,L], max_len [1])."""
        B, T, _ = enc_states.shape
        enc_lens = torch.round(T * wav_len.to(enc_states.device)).int()
        cw = cb = None
        if self.ctc_weight > 0:
            cw, cb = self.ctc_fc.w.weight, self.ctc_fc.w.bias
        tok, ln, sc, lp, mxl, steps = native.beam_search(self._handle(), self.config(T), enc_states.contiguous(),
                                                         enc_lens, cw, cb)
        return tok, ln, sc, lp, mxl

    def _forward_each(self, items, ratios=None):
        """``forward`` batch by batch under each batch's own decode ratios (the fallback of ``forward_group``)."""
        keep = (self.min_decode_ratio, self.max_decode_ratio)
        out = []
        try:
            for i, (enc, wl) in enumerate(items):
                if ratios is not None:
                    self.min_decode_ratio, self.max_decode_ratio = ratios[i]
                out.append(self.forward(enc, wl))
        finally:
            self.min_decode_ratio, self.max_decode_ratio = keep
        return out

    @torch.no_grad()
    def forward_group(self, items, ratios=None):
        """Several independent batches in ONE device search.

        ``items``: list of (enc_states [B_g,T_g,d], wav_len [B_g]); ``ratios`` (optional): per batch
        (min_decode_ratio, max_decode_ratio), default the searcher's own.  Returns the list of ``forward`` results,
        one per batch (with ``return_topk`` the padded top-k tensors of each batch).  Every batch keeps its own semantics -- its own padded length T_g, hence its own step limits
        int(T_g * ratio) (seq2seq.py:1336-1338) and its own length normaliser for ``best_lens`` -- but the decoder
        step runs over the rows of all batches at once (csrc/search.hip, ``utt_max_steps``): with recipe-sized
        batches the per-step GEMMs otherwise see a few hundred rows and cannot fill the chip."""
        if self.ctc_window_size and len(items) > 1:
            # the CTC attention window takes its frame range over the whole batch of a search (ctc.py:189-200): the batches keep their
            # own searches -- what the reference, which has no grouped search, does with them
            return self._forward_each(items, ratios)
        if len(items) == 1 and ratios is None:
            return [self.forward(*items[0])]
        dev = items[0][0].device
        ratios = ratios or [(self.min_decode_ratio, self.max_decode_ratio)] * len(items)
        Bs = [e.shape[0] for e, _ in items]
        Tmax, d = max(e.shape[1] for e, _ in items), items[0][0].shape[2]
        enc = torch.zeros(sum(Bs), Tmax, d, dtype=torch.float32, device=dev)  # frames past enc_len are never read
        lens, mins, maxs, off = [], [], [], 0
        for (e, wl), (r_min, r_max), B in zip(items, ratios, Bs):
            T = e.shape[1]
            enc[off: off + B, :T] = e
            lens.append(torch.round(T * wl.to(dev)).int())
            mins += [int(T * r_min)] * B
            maxs += [int(T * r_max)] * B
            off += B
        enc_lens = torch.cat(lens)
        limits = torch.tensor([mins, maxs], dtype=torch.int32).to(dev, non_blocking=True)
        cfg = self.config(Tmax)
        cfg.min_steps, cfg.max_steps = min(mins), max(maxs)
        cw = cb = None
        if self.ctc_weight > 0:
            cw, cb = self.ctc_fc.w.weight, self.ctc_fc.w.bias
        tok, ln, sc, lp, _, _, longest = native.beam_search(self._handle(), cfg, enc, enc_lens, cw, cb,
                                                            utt_min_steps=limits[0].contiguous(),
                                                            utt_max_steps=limits[1].contiguous(), want_longest=True)
        n, L = tok.shape
        if self.return_topk:  # every batch as forward() returns it: padded [B_g, topk, max_len_g] tensors (seq2seq.py:1712-1713)
            K, longest_h = self.topk, longest.cpu()  # (the call's one sync; rows are utterance-major, topk per utterance)
            out, off = [], 0
            for B in Bs:
                max_len = max(int(longest_h[off: off + B].max()), 1)  # this batch's pad width (seq2seq.py:1461)
                rows = slice(off * K, (off + B) * K)
                out.append((tok[rows, :max_len].reshape(B, K, max_len).long(), ln[rows].float().reshape(B, K) / max_len,
                            sc[rows].reshape(B, K), lp[rows, :max_len].reshape(B, K, max_len)))
                off += B
            return out
        packed = torch.cat([tok.reshape(-1), ln, longest]).cpu()  # one device->host copy
        tok_h, ln_h, longest_h = packed[: n * L].reshape(n, L), packed[n * L: n * L + n], packed[n * L + n:]
        out, off = [], 0
        for B in Bs:
            max_len = max(int(longest_h[off: off + B].max()), 1)  # this batch's pad width (seq2seq.py:1461)
            hyps = [tok_h[b, : int(ln_h[b])].tolist() for b in range(off, off + B)]
            out.append((hyps, (ln_h[off: off + B].float() / max_len).to(dev), sc[off: off + B],
                        lp[off: off + B, :max_len]))
            off += B
        return out

    @torch.no_grad()
    def forward(self, enc_states, wav_len):
        tok, ln, sc, lp, mxl = self.search_device(enc_states, wav_len)
        if self.return_topk:  # padded tensors [B,topk,max_len] (seq2seq.py:1712-1713)
            B, K, max_len = enc_states.shape[0], self.topk, max(int(mxl.cpu()), 1)
            return (tok[:, :max_len].reshape(B, K, max_len).long(), ln.float().reshape(B, K) / max_len,
                    sc.reshape(B, K), lp[:, :max_len].reshape(B, K, max_len))
        B, L = tok.shape
        packed = torch.cat([tok.reshape(-1), ln, mxl]).cpu()  # one device->host copy (and the call's only sync)
        tok_h, ln_h, max_len = packed[: B * L].reshape(B, L), packed[B * L: B * L + B], max(int(packed[-1]), 1)
        hyps = [tok_h[b, : int(ln_h[b])].tolist() for b in range(B)]
        best_lens = ln_h.float() / max_len  # SpeechBrain relative length (seq2seq.py:1461)
        return hyps, best_lens.to(enc_states.device), sc, lp[:, :max_len]


class S2SWhisperBeamSearcher(_WhisperPrompting, S2STransformerBeamSearcher):
    """seq2seq.py:1937-2206 (on S2SBeamSearcher, :711-1749): beam search of a
    ``speechbrain_amd.integrations.huggingface.whisper.Whisper``.  The initial tokens prime the device-side KV cache of
    every hypothesis (``reset_mem``), search step s runs at decoder position sample_begin-1+s, the suppression masks
    are additive -inf biases on the logits, log-probs are ``log_softmax(logits) / temperature`` (:2189-2192), the search
    stops after ``max_attn_tokens - sample_begin`` steps (:2196-2201) -- all inside one C-ABI call
    (sbk_beam_search_f32 with the ``prompt`` fields of sbk_search_config).

    forward(enc_states, wav_len) -> (hyps, best_lens, best_scores, best_log_probs) as the reference; ``no_speech_probs``
    is set as in the reference.  A ``scorer`` (CTC / LM) is not supported with Whisper."""

    def __init__(self, module, temperature=1.0, use_kv_cache=True, suppress_blank=True, suppress_tokens="-1",
                 sample_len=None, prefix=None, prompt=None, **kwargs):
        model = module[0]
        if kwargs.get("scorer") is not None:
            raise NotImplementedError("S2SWhisperBeamSearcher with a scorer")
        kwargs.setdefault("min_decode_ratio", 0.0)
        kwargs.setdefault("max_decode_ratio", 1.0)
        S2STransformerBeamSearcher.__init__(self, modules=[model, None], temperature=temperature, bos_index=model.bos,
                                            eos_index=model.eos, **kwargs)
        self.use_kv_cache = use_kv_cache
        self.suppress_blank, self.suppress_tokens = suppress_blank, suppress_tokens
        self.prefix, self.prompt = prefix, prompt
        cfg = model.model.config
        self.max_attn_tokens = cfg.get("max_length", cfg.get("max_target_positions", 448))
        self.sample_len = sample_len or self.max_attn_tokens // 2
        self.no_speech_probs, self.lang_tokens = None, None
        self.overlap_ctc = 0
        self._refresh()

    def _handle(self):
        return self.model.decoder_handle()

    @torch.no_grad()
    def search_device(self, enc_states, wav_len=None):
        enc = enc_states.float().contiguous()
        B, T, _ = enc.shape
        dev = enc.device
        P = self.sample_begin
        prompt = torch.tensor([list(self.initial_tokens)] * B, dtype=torch.int32)
        if self.lang_tokens is not None:  # :2113-2119: language token right after <|startoftranscript|>
            lt = torch.as_tensor(self.lang_tokens).to(torch.int32).cpu().reshape(-1)
            if lt.numel() == B * self.beam_size and self.beam_size > 1:
                lt = lt[:: self.beam_size]  # the reference's memory holds one row per hypothesis (batch x beam)
            prompt[:, self.initial_tokens.index(self.model.bos) + 1] = lt  # (1 or B entries)
        handle = self._handle()
        always, first = self._masks(handle.W.vocab, dev)
        cfg = self.config(T)
        # S2SBeamSearcher.forward runs range(max_decode_steps) and leaves after the step that makes the hypotheses
        # max_attn_tokens - sample_begin long (:1664-1696, :2196-2201): at least one step, at most that many
        cfg.max_steps = max(1, min(cfg.max_steps, self.max_attn_tokens - P)) if cfg.max_steps > 0 else 0
        cfg.min_steps = min(cfg.min_steps, cfg.max_steps)
        cfg.graph_mode, cfg.overlap_ctc = 0, 0
        self._prompt_dev = prompt.to(dev)
        probe = torch.zeros(B, dtype=torch.float32, device=dev)
        cfg.prompt, cfg.prompt_len = self._prompt_dev.data_ptr(), P
        cfg.temperature_post = 1
        cfg.logit_bias = always.data_ptr() if always is not None else None
        cfg.first_bias = first.data_ptr() if first is not None else None
        cfg.probe_pos, cfg.probe_token = self.initial_tokens.index(self.model.bos), self.model.no_speech
        cfg.out_probe = probe.data_ptr()
        full = torch.full((B,), T, dtype=torch.int32, device=dev)  # the Whisper decoder attends to every encoder frame
        tok, ln, sc, lp, mxl, steps = native.beam_search(handle, cfg, enc, full)
        self.no_speech_probs = probe.cpu().tolist()
        return tok, ln, sc, lp, mxl

    def forward(self, enc_states, wav_len=None):
        return S2STransformerBeamSearcher.forward(self, enc_states, wav_len)

    def forward_group(self, items, ratios=None):
        # a token prompt primes every hypothesis' cache at its own decoder positions: the batches keep their own searches
        return self._forward_each(items, ratios)
