"""Known answers the reference's own tests/doctests hold for this path, replayed on the oracle
(SURVEY.md section 8c): tests/unittests/test_features.py:57-118, doctests in
processing/features.py:365-367,1580-1599, Transformer.py:1052-1056, dataio/dataio.py:838-844."""
import torch

from oracle import sb_oracle as O


def test_filterbank_silence_is_minus_100_db():
    # test_features.py:64-68 (amin -> -100 dB) and :70-74 (top_db on a 1x1x1 input)
    out = O.fbank(torch.zeros(10, 16000), O.FbankCfg())
    assert out.shape == (10, 101, 40)  # lobes/features.py:90-95 doctest shape
    assert torch.equal(out, torch.ones_like(out) * -100)


def test_fbank_batch_invariance():
    # test_features.py:76-84: independent computation == batched computation (sum |d| < 8e-5 per item)
    g = torch.Generator().manual_seed(0)
    a, b = torch.rand(1, 8000, generator=g), torch.rand(1, 8000, generator=g) * 0.1
    both = O.fbank(torch.cat([a, b]), O.FbankCfg())
    assert float((O.fbank(a, O.FbankCfg())[0] - both[0]).abs().max()) <= 8e-5
    assert float((O.fbank(b, O.FbankCfg())[0] - both[1]).abs().max()) <= 8e-5


def test_input_normalization_known_answer():
    # test_features.py:110-118: stats (mean 2, std 1) from the two valid frames -> [-1, 1, -2, -2, -2]
    x = torch.tensor([1.0, 3.0, 0.0, 0.0, 0.0]).view(1, -1, 1)
    out = O.input_norm_global(x, torch.tensor([2.0]), torch.tensor([1.0])).squeeze()
    assert torch.equal(out, torch.tensor([-1.0, 1.0, -2.0, -2.0, -2.0]))


def test_make_padding_mask_doctest():
    # processing/features.py:1580-1599
    m = O.padding_mask(4, torch.tensor([1.0, 0.75, 0.5]))
    assert m.tolist() == [[True] * 4, [True, True, True, False], [True, True, False, False]]


def test_length_to_mask_doctest():
    # dataio/dataio.py:838-844
    assert O.length_to_mask(torch.tensor([1, 2, 3])).int().tolist() == [[1, 0, 0], [1, 1, 0], [1, 1, 1]]


def test_relpos_table_symmetry():
    # pos[T-1-r] == pos[T-1+r]; centre row is [0,1,0,1,...]
    T, d = 9, 8
    pe = O.relpos_table(T, d)
    assert pe.shape == (2 * T - 1, d)
    for r in range(T):
        assert torch.equal(pe[T - 1 - r], pe[T - 1 + r])
    assert pe[T - 1].tolist() == [0.0, 1.0] * (d // 2)


def test_rel_shift_indexing():
    # attention.py:537-553: after the shift, score(i,j) reads column T-1-i+j
    T = 5
    x = torch.arange(T * (2 * T - 1), dtype=torch.float32).view(1, 1, T, 2 * T - 1)
    y = O.rel_shift(x)
    for i in range(T):
        for j in range(T):
            assert y[0, 0, i, j] == x[0, 0, i, T - 1 - i + j]
