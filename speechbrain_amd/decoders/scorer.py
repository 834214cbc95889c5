"""speechbrain.decoders.scorer mirror for the ASR recipe: ScorerBuilder + CTCScorer + TransformerLMScorer.

Two ways to use them.  ``S2STransformerBeamSearcher(scorer=ScorerBuilder(...))`` reads them as configuration: the CTC prefix
scoring and the LM steps are fused into the device-side search (csrc/ctc_prefix.hip, csrc/search.hip).  A searcher written
against the reference's own step protocol (scorer.py:1221-1315: ``reset_scorer_mem`` once per batch, then ``score`` and
``permute_scorer_mem`` once per decoding step) calls the same methods here: ``CTCScorer`` runs the library's prefix scorer
through its per-step entry points (``sbk_ctc_scorer_*``), ``TransformerLMScorer`` re-runs the LM over the prefix like the
reference does (``sbk_lm_prefix_f32``).
"""
import torch


class BaseScorerInterface:
    """scorer.py:24-105: score(inp_tokens, memory, candidates, attn) -> (scores [n_bh, V], memory); permute_mem; reset_mem."""

    def score(self, inp_tokens, memory, candidates, attn):
        raise NotImplementedError

    def permute_mem(self, memory, index):
        pass

    def reset_mem(self, x, enc_lens):
        pass


class CTCScorer(BaseScorerInterface):
    """scorer.py:108-255: ctc_fc = the CTC output Linear; blank_index / eos_index as in the recipe;
    ``ctc_window_size`` > 0 restricts the scored frames of a step to a window around the attention peaks
    (ctc.py:189-200; csrc/search.hip keeps the running peaks of the last decoder layer's cross-attention)."""

    minus_inf = -1e20  # CTCPrefixScore.minus_inf (ctc.py:53)

    def __init__(self, ctc_fc, blank_index, eos_index, ctc_window_size=0):
        if ctc_window_size < 0:
            raise ValueError("ctc_window_size must be >= 0")
        self.ctc_fc, self.blank_index, self.eos_index = ctc_fc, blank_index, eos_index
        self.ctc_window_size = ctc_window_size
        self.ctc_score = None

    # ---- the reference's per-step protocol (scorer.py:183-255)
    def reset_mem(self, x, enc_lens):
        """x [B,T,d] encoder states, enc_lens relative lengths: log_softmax(ctc_fc(x)) and a fresh prefix scorer (scorer.py:239-255)."""
        from speechbrain_amd import native

        logp = native.log_softmax(self.ctc_fc(x))
        lens = torch.round(enc_lens.to(x.device).float() * x.shape[1]).to(torch.int32)  # ctc.py:58 (relative -> frames)
        self.ctc_score = native.CTCStepScorer(logp, lens, self.blank_index, self.eos_index, self.ctc_window_size)
        return None

    def score(self, inp_tokens, memory, candidates, attn):
        """(psi - psi_prev [n_bh, V], memory): CTCPrefixScore.forward_step (ctc.py:79-262).  ``candidates`` [n_bh, k]: only those
        tokens (and <eos>) keep a CTC score, every other entry is minus_inf - psi_prev as in the reference (ctc.py:246-254).
        ``attn`` [n_bh, T] with ctc_window_size > 0: the scored frames follow the attention peaks (ctc.py:189-200)."""
        step = 0 if memory is None else memory[0] + 1
        win = None
        if self.ctc_window_size > 0 and attn is not None:
            peak = attn.argmax(dim=1)
            win = torch.stack([peak.min(), peak.max()]).to(torch.int32)
        sc = self.ctc_score.score(inp_tokens, step, win)
        if candidates is not None:
            keep = torch.zeros_like(sc, dtype=torch.bool)
            keep.scatter_(1, candidates.long(), True)
            keep[:, self.eos_index] = True  # (psi[eos] is assigned for every hypothesis, candidates or not: ctc.py:256-259)
            # a token outside the candidates: psi = minus_inf (ctc.py:247-248), so the score is minus_inf - psi_prev; psi_prev is
            # what the blank column of the full score already carries (psi[blank] = minus_inf for every hypothesis)
            if self.eos_index != self.blank_index:
                sc = torch.where(keep, sc, sc[:, self.blank_index: self.blank_index + 1].expand_as(sc))
            else:
                sc = torch.where(keep, sc, torch.full_like(sc, self.minus_inf))
        return sc, (step, inp_tokens, win)

    def permute_mem(self, memory, index):
        """index [B, beam]: flat candidate ids over beam * V of each utterance, as the beam update selects them (ctc.py:243-295)."""
        step, inp_tokens, win = memory
        B, beam = index.shape
        V = self.ctc_score.V
        off = (torch.arange(B, device=index.device) * beam).unsqueeze(1)
        parent = (torch.div(index, V, rounding_mode="floor") + off).reshape(-1)
        token = (index % V).reshape(-1)
        self.ctc_score.permute(parent, token, inp_tokens, step, win)
        return (step, None, None)


class TransformerLMScorer(BaseScorerInterface):
    """scorer.py:413-577: language_model = a TransformerLM; temperature divides its logits before the
    log-softmax.  The reference re-runs the LM over the whole prefix every step (:538-543); the device
    search keeps a per-hypothesis K/V cache instead (csrc/search.hip:lm_step)."""

    def __init__(self, language_model, temperature=1.0):
        self.lm = language_model
        self.lm.eval()
        self.temperature = temperature
        self._handle = None

    # ---- the reference's per-step protocol (scorer.py:507-560): the prefix is the memory, the LM runs over all of it
    def score(self, inp_tokens, memory, candidates, attn):
        from speechbrain_amd import native

        if memory is None:
            memory = torch.empty(inp_tokens.shape[0], 0, dtype=torch.int32, device=inp_tokens.device)
        memory = torch.cat([memory, inp_tokens.to(torch.int32).unsqueeze(1)], dim=-1).contiguous()
        if self._handle is None or self._handle.stale(self.lm):
            self._handle = native.LMHandle(self.lm)
        logits = native.lm_prefix(self._handle, memory)[:, -1, :].contiguous()
        return native.log_softmax(logits, temperature=self.temperature), memory

    def permute_mem(self, memory, index):
        """index [n_bh]: the row of the previous beam every new hypothesis extends (scorer.py:545-560)."""
        return torch.index_select(memory, 0, index.reshape(-1).long())

    def reset_mem(self, x, enc_lens):
        return None


_KNOWN = ("ctc", "rnnlm", "transformerlm", "kenlm", "coverage", "length", "huggingfacelm", "basescorerinterface")


class ScorerBuilder:
    """scorer.py:1075-1315.  Supported compositions: full_scorers drawn from {CTCScorer,
    TransformerLMScorer} -- the recipe's ``valid_search`` ([ctc]) and ``test_search``
    ([transformerlm, ctc]) (conformer_large.yaml:215-239) -- and CTCScorer as the one partial scorer
    (``partial_scorers=[ctc]``: only the best int(beam_size * scorer_beam_scale) tokens of every hypothesis get a
    CTC score, :1287-1300)."""

    def __init__(self, weights=dict(), full_scorers=list(), partial_scorers=list(), scorer_beam_scale=2):
        assert len(weights) == len(full_scorers) + len(partial_scorers), "Weights and scorers are not matched."
        self.scorer_beam_scale = scorer_beam_scale
        name = lambda impl: impl.__class__.__name__.lower().split("scorer")[0]  # noqa: E731
        self.weights = {**dict.fromkeys(_KNOWN, 0.0), **weights}
        self.full_scorers = {name(s): s for s in full_scorers}
        self.partial_scorers = {name(s): s for s in partial_scorers}
        unsupported = [k for k in self.full_scorers if k not in ("ctc", "transformerlm")]
        unsupported += [k for k in self.partial_scorers if k != "ctc"]
        if unsupported or ("ctc" in self.full_scorers and "ctc" in self.partial_scorers):
            raise NotImplementedError(
                f"scorers {unsupported}: the device search fuses the CTC scorer (full or partial) and the full "
                "TransformerLM scorer; RNNLM / KenLM / coverage / length scorers are not implemented")
        if not 0.0 <= self.weights["ctc"] <= 1.0:
            raise ValueError("ctc_weight should not > 1.0 and < 0.0")

    # ---- the reference's per-step protocol (scorer.py:1221-1315), for searchers that drive the scorers themselves
    def score(self, inp_tokens, memory, attn, log_probs, beam_size):
        """log_probs [n_bh, V] (updated in place and returned) += weight * score of every full scorer, then of every partial
        scorer on the int(beam_size * scorer_beam_scale) best tokens of each hypothesis (scorer.py:1221-1266)."""
        new_memory = dict()
        for k, impl in self.full_scorers.items():
            if k == "ctc":
                log_probs[:, impl.blank_index] = impl.minus_inf  # block blank token if CTC is used (:1247-1250)
            score, new_memory[k] = impl.score(inp_tokens, memory[k], None, attn)
            log_probs += score * self.weights[k]
        num_candidates = int(beam_size * self.scorer_beam_scale)
        num_candidates = max(1, min(num_candidates, log_probs.shape[-1]))
        candidates = log_probs.topk(num_candidates, dim=-1).indices
        for k, impl in self.partial_scorers.items():
            score, new_memory[k] = impl.score(inp_tokens, memory[k], candidates, attn)
            log_probs += score * self.weights[k]
        return log_probs, new_memory

    def permute_scorer_mem(self, memory, index, candidates):
        """index [n_bh]: previous path of every new hypothesis; candidates [B, beam]: flat top-k ids (scorer.py:1268-1296)."""
        for k, impl in self.full_scorers.items():
            memory[k] = impl.permute_mem(memory[k], candidates if k == "ctc" else index)
        for k, impl in self.partial_scorers.items():
            memory[k] = impl.permute_mem(memory[k], candidates)
        return memory

    def reset_scorer_mem(self, x, enc_lens):
        """scorer.py:1298-1315."""
        return {k: impl.reset_mem(x, enc_lens) for k, impl in {**self.full_scorers, **self.partial_scorers}.items()}
