"""ErrorRateStats / edit distance (speechbrain_amd/utils/{metric_stats,edit_distance}.py) against the
REFERENCE's outputs (tests/golden/wer.npz, written by oracle/make_golden.py --wer-only from
/root/reference/speechbrain/utils/metric_stats.py:206) and the reference's doctest known answers."""
import os

import numpy as np
import torch

from speechbrain_amd.utils.edit_distance import alignment, count_ops, op_table, wer_details_for_batch
from speechbrain_amd.utils.metric_stats import ErrorRateStats, token_error_rate

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wer.npz")


def test_reference_doctest_values():
    # utils/edit_distance.py:176-183
    assert op_table([1, 2, 3], [1, 2, 4]) == [list("=III"), list("D=II"), list("DD=I"), list("DDDS")]
    # :264-271, :329-336
    table = [list("IIII"), list("D=II"), list("DD=I"), list("DDDS")]
    assert alignment(table) == [("=", 0, 0), ("=", 1, 1), ("S", 2, 2)]
    assert dict(count_ops(table)) == {"substitutions": 1}
    # :391-404
    d = wer_details_for_batch(["utt1"], [["a", "b", "c"]], [["a", "b", "d"]])
    assert abs(d[0]["WER"] - 33.3333) < 1e-3
    # utils/metric_stats.py:236-253
    i2l = {0: "a", 1: "b"}
    cer = ErrorRateStats()
    cer.append(ids=["utterance1"], predict=torch.tensor([[0, 1, 1]]), target=torch.tensor([[0, 1, 0]]),
               target_len=torch.ones(1), ind2lab=lambda batch: [[i2l[int(x)] for x in seq] for seq in batch])
    s = cer.summarize()
    assert abs(s["WER"] - 33.3333) < 1e-3 and (s["insertions"], s["deletions"], s["substitutions"]) == (0, 0, 1)


def test_against_reference_golden():
    g = np.load(GOLD)
    refs = [[int(v) for v in row if v >= 0] for row in g["refs"]]
    hyps = [[int(v) for v in row if v >= 0] for row in g["hyps"]]
    stats = ErrorRateStats()
    stats.append([f"u{i}" for i in range(len(refs))], [[str(t) for t in h] for h in hyps],
                 [[str(t) for t in r] for r in refs])
    got = np.array([[d["insertions"], d["deletions"], d["substitutions"]] for d in stats.scores])
    assert np.array_equal(got, g["ins_del_sub"])
    assert np.allclose([d["WER"] for d in stats.scores], g["utt_wer"], rtol=0, atol=1e-9)
    assert ["".join(op for op, _, _ in d["alignment"]) for d in stats.scores] == [str(s) for s in g["ops"]]
    summ = stats.summarize()
    for k, v in zip(g["summary_keys"], g["summary_vals"]):
        assert abs(float(summ[str(k)]) - float(v)) < 1e-9, k
    assert abs(token_error_rate(hyps, refs)["WER"] - float(summ["WER"])) < 1e-12
