"""speechbrain.utils.metric_stats mirror: ErrorRateStats (utils/metric_stats.py:206-385), the routine the
recipes score WER with (`within 0.1 abs of the reference` is a statement about this number).  Host logic."""
from speechbrain_amd.utils.data_utils import undo_padding
from speechbrain_amd.utils.edit_distance import _same, wer_details_for_batch, wer_summary


def merge_char(sequences, space="_"):
    """dataio/dataio.py merge_char: character tokens -> words, split on `space`."""
    return ["".join(seq).split(space) for seq in sequences]


def split_word(sequences, space="_"):
    """dataio/dataio.py split_word: word tokens -> characters with `space` between words."""
    return [list(space.join(seq)) for seq in sequences]


class ErrorRateStats:
    def __init__(self, merge_tokens=False, split_tokens=False, space_token="_", keep_values=True,
                 extract_concepts_values=False, tag_in="", tag_out="", equality_comparator=_same):
        if extract_concepts_values:
            raise NotImplementedError("concept/value extraction (SLU scoring) is not on the ASR path")
        self.merge_tokens, self.split_tokens, self.space_token = merge_tokens, split_tokens, space_token
        self.equality_comparator = equality_comparator
        self.clear()

    def clear(self):
        self.scores, self.ids, self.summary = [], [], {}

    def append(self, ids, predict, target, predict_len=None, target_len=None, ind2lab=None):
        self.ids.extend(ids)
        if predict_len is not None:
            predict = undo_padding(predict, predict_len)
        if target_len is not None:
            target = undo_padding(target, target_len)
        if ind2lab is not None:
            predict, target = ind2lab(predict), ind2lab(target)
        if self.merge_tokens:
            predict, target = merge_char(predict, self.space_token), merge_char(target, self.space_token)
        if self.split_tokens:
            predict, target = split_word(predict, self.space_token), split_word(target, self.space_token)
        self.scores.extend(wer_details_for_batch(ids, target, predict, compute_alignments=True,
                                                 equality_comparator=self.equality_comparator))

    def summarize(self, field=None):
        self.summary = wer_summary(self.scores)
        self.summary["error_rate"] = self.summary["WER"]
        return self.summary[field] if field is not None else self.summary


def token_error_rate(hyps, refs):
    """WER (%) of token-id lists `hyps` against `refs` -- what bench.py reports between the HIP path and the
    oracle's tokens for the same utterances."""
    stats = ErrorRateStats()
    stats.append(list(range(len(refs))), [list(map(int, h)) for h in hyps], [list(map(int, r)) for r in refs])
    return stats.summarize()
