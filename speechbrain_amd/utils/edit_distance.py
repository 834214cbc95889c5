"""speechbrain.utils.edit_distance mirror: the WER arithmetic behind ErrorRateStats (utils/edit_distance.py:146-633).

Host logic (integer dynamic programming over token lists; nothing here is on the GPU path).  The tie-breaking
order of the reference -- Kaldi compute-wer's: in a tie insertion wins over deletion wins over substitution
(edit_distance.py:213-229) -- decides how an edit distance splits into insertions / deletions / substitutions,
so it is reproduced exactly; `tests/test_wer.py` pins it against the reference's outputs.
"""
import collections
from typing import Callable, Dict, List, Sequence

EDIT_SYMBOLS = {"eq": "=", "ins": "I", "del": "D", "sub": "S"}  # the reference's alignment alphabet


def _same(a, b):
    return a == b


def op_table(a: Sequence, b: Sequence, equality_comparator: Callable = _same) -> List[List[str]]:
    """Edit-operation table [len(a)+1][len(b)+1] from reference `a` to hypothesis `b`."""
    na, nb = len(a), len(b)
    ops = [[EDIT_SYMBOLS["eq"]] * (nb + 1) for _ in range(na + 1)]
    for i in range(1, na + 1):
        ops[i][0] = EDIT_SYMBOLS["del"]
    for j in range(1, nb + 1):
        ops[0][j] = EDIT_SYMBOLS["ins"]
    above = list(range(nb + 1))  # costs of the previous row
    for i in range(1, na + 1):
        row = [i] + [0] * nb
        ai, op_row = a[i - 1], ops[i]
        for j in range(1, nb + 1):
            differs = 0 if equality_comparator(ai, b[j - 1]) else 1
            c_sub, c_del, c_ins = above[j - 1] + differs, above[j] + 1, row[j - 1] + 1
            if c_sub < c_ins and c_sub < c_del:
                row[j] = c_sub
                if differs:
                    op_row[j] = EDIT_SYMBOLS["sub"]
            elif c_del < c_ins:
                row[j], op_row[j] = c_del, EDIT_SYMBOLS["del"]
            else:
                row[j], op_row[j] = c_ins, EDIT_SYMBOLS["ins"]
        above = row
    return ops


def _walk(table):
    """Yield (op, i, j) from the end of the table back to its origin (indices AFTER the step)."""
    i, j = len(table) - 1, len(table[0]) - 1
    while i or j:
        op = EDIT_SYMBOLS["ins"] if i == 0 else EDIT_SYMBOLS["del"] if j == 0 else table[i][j]
        if op == EDIT_SYMBOLS["ins"]:
            j -= 1
            yield op, None, j
        elif op == EDIT_SYMBOLS["del"]:
            i -= 1
            yield op, i, None
        else:
            i, j = i - 1, j - 1
            yield op, i, j


def alignment(table):
    """[(op, index into a or None, index into b or None)] in forward order."""
    return list(_walk(table))[::-1]


def count_ops(table):
    names = {EDIT_SYMBOLS["ins"]: "insertions", EDIT_SYMBOLS["del"]: "deletions", EDIT_SYMBOLS["sub"]: "substitutions"}
    edits = collections.Counter()
    for op, _, _ in _walk(table):
        if op in names:
            edits[names[op]] += 1
    return edits


def wer_details_for_batch(ids, refs, hyps, compute_alignments=False, equality_comparator: Callable = _same):
    """Per-utterance details in the reference's schema (edit_distance.py:372-557), scoring mode "strict"."""
    out = []
    for key, ref, hyp in zip(ids, refs, hyps):
        ref, hyp = list(ref), list(hyp)
        table = op_table(ref, hyp, equality_comparator)
        ops = count_ops(table)
        n_edits = sum(ops.values())
        n_ref = 0 if (not ref or ref[0] == "") else len(ref)  # "" outputs count as empty (edit_distance.py:529-533)
        d = {"key": key, "scored": True, "hyp_absent": False, "hyp_empty": len(hyp) == 0, "num_edits": n_edits,
             "num_ref_tokens": n_ref, "WER": 100.0 * n_edits / max(1, n_ref),
             "insertions": ops["insertions"], "deletions": ops["deletions"], "substitutions": ops["substitutions"],
             "alignment": alignment(table) if compute_alignments else None,
             "ref_tokens": ref if compute_alignments else None, "hyp_tokens": hyp if compute_alignments else None}
        out.append(d)
    return out


def wer_summary(details_by_utterance) -> Dict:
    """edit_distance.py:560-632."""
    tot = collections.Counter()
    for d in details_by_utterance:
        tot["num_ref_sents"] += 1
        if d["scored"]:
            tot["num_scored_sents"] += 1
            tot["num_scored_tokens"] += d["num_ref_tokens"]
            for k in ("insertions", "deletions", "substitutions", "num_edits"):
                tot[k] += d[k]
            tot["num_erroneous_sents"] += 1 if d["num_edits"] > 0 else 0
        tot["num_absent_sents"] += 1 if d["hyp_absent"] else 0
    toks, sents = tot["num_scored_tokens"], tot["num_scored_sents"]
    summary = {"WER": 100.0 * tot["num_edits"] / toks if toks else 0.0,
               "SER": 100.0 * tot["num_erroneous_sents"] / sents if sents else 0.0}
    for k in ("num_edits", "num_scored_tokens", "num_erroneous_sents", "num_scored_sents", "num_absent_sents",
              "num_ref_sents", "insertions", "deletions", "substitutions"):
        summary[k] = tot[k]
    return summary
