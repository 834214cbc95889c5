"""speechbrain.decoders mirror (ASR searchers + CTC / TransformerLM scorers)."""
from speechbrain_amd.decoders.scorer import CTCScorer, ScorerBuilder, TransformerLMScorer  # noqa: F401
from speechbrain_amd.decoders.seq2seq import (S2STransformerBeamSearcher, S2STransformerGreedySearcher,  # noqa: F401
                                                S2SWhisperBeamSearcher, S2SWhisperGreedySearcher)
