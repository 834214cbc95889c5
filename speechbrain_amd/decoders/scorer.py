"""speechbrain.decoders.scorer mirror for the ASR recipe: ScorerBuilder + CTCScorer + TransformerLMScorer.

The reference scorer objects carry Python-side state and are called once per decoding step
(scorer.py:1221-1315, :108-255).  Here they are configuration holders: the CTC prefix scoring
itself is fused into the device-side search (csrc/ctc_prefix.hip), driven by
``S2STransformerBeamSearcher``.
"""


class BaseScorerInterface:
    pass


class CTCScorer(BaseScorerInterface):
    """scorer.py:108-255: ctc_fc = the CTC output Linear; blank_index / eos_index as in the recipe;
    ``ctc_window_size`` > 0 restricts the scored frames of a step to a window around the attention peaks
    (ctc.py:189-200; csrc/search.hip keeps the running peaks of the last decoder layer's cross-attention)."""

    def __init__(self, ctc_fc, blank_index, eos_index, ctc_window_size=0):
        if ctc_window_size < 0:
            raise ValueError("ctc_window_size must be >= 0")
        self.ctc_fc, self.blank_index, self.eos_index = ctc_fc, blank_index, eos_index
        self.ctc_window_size = ctc_window_size


class TransformerLMScorer(BaseScorerInterface):
    """scorer.py:413-577: language_model = a TransformerLM; temperature divides its logits before the
    log-softmax.  The reference re-runs the LM over the whole prefix every step (:538-543); the device
    search keeps a per-hypothesis K/V cache instead (csrc/search.hip:lm_step)."""

    def __init__(self, language_model, temperature=1.0):
        self.lm = language_model
        self.lm.eval()
        self.temperature = temperature


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
