"""CPU tier: the reference's own tests of CrossValidation and HoldOut (/root/reference/tests/dataset/
crossvalidation_test.py, holdout_test.py) re-typed against this package on the same table
(util_test.generate_normal_data(10000) = the golden file's train10k)."""
import numpy as np
import pandas as pd
import pytest

import pybnesian_amd as pbn
from helpers import frame

SIZE = 10000


@pytest.fixture(scope="module")
def df(golden, ensure_built):
    return frame(golden["train10k"])


def null_table(df):
    np.random.seed(0)
    out = df.copy()
    for c in "abcd":
        out.loc[out.index[np.random.randint(0, SIZE, size=100)], c] = np.nan
    return out


def test_cv_disjoint_indices_fold_seed(df):   # crossvalidation_test.py:10-70
    cv = pbn.CrossValidation(df)
    pairs = list(zip(cv, cv.indices()))
    assert len(pairs) == 10
    for i, ((train_df, test_df), (tr, te)) in enumerate(pairs):
        assert np.all(np.sort(np.hstack((tr, te))) == np.arange(SIZE))
        assert np.all(train_df.to_pandas().to_numpy() == df.iloc[tr, :].to_numpy())
        assert np.all(test_df.to_pandas().to_numpy() == df.iloc[te, :].to_numpy())
        assert np.setdiff1d(tr, te).shape == tr.shape and np.setdiff1d(te, tr).shape == te.shape
        ftr, fte = cv.fold(i)
        assert ftr.equals(train_df) and fte.equals(test_df)
    cv0, cv0b, cv1 = pbn.CrossValidation(df, seed=0), pbn.CrossValidation(df, seed=0), pbn.CrossValidation(df, seed=1)
    for (a, b), (a2, b2), (a3, b3) in zip(cv0, cv0b, cv1):
        assert a.equals(a2) and b.equals(b2) and not a.equals(a3) and not b.equals(b3)
    cv5 = pbn.CrossValidation(df, 5)
    assert len(list(cv5)) == 5 and len(list(cv5.indices())) == 5


def test_cv_loc_and_null(df):   # crossvalidation_test.py:72-130
    cv = pbn.CrossValidation(df)
    for sel, names in (("a", ["a"]), (1, ["b"]), (["b", "d"], ["b", "d"]), ([0, 2], ["a", "c"])):
        for train_df, test_df in cv.loc(sel):
            assert train_df.schema.names == names and test_df.schema.names == names
    dn = null_table(df)
    non_null = dn.dropna()
    cvn = pbn.CrossValidation(dn)
    for (train_df, test_df), (tr, te) in zip(cvn, cvn.indices()):
        assert non_null.shape[0] == train_df.num_rows + test_df.num_rows
        assert np.all(np.sort(np.hstack((tr, te))) == np.sort(non_null.index.to_numpy()))
        assert np.all(train_df.to_pandas().to_numpy() == dn.iloc[tr, :].to_numpy())
    cvi = pbn.CrossValidation(dn, include_null=True)
    for train_df, test_df in cvi:
        assert train_df.num_rows + test_df.num_rows == SIZE


def test_holdout(df):   # holdout_test.py:11-90
    for ratio in (0.2, 0.3):
        hold = pbn.HoldOut(df, test_ratio=ratio)
        tr, te = hold.training_data(), hold.test_data()
        assert tr.num_rows + te.num_rows == SIZE
        assert tr.num_rows == round((1 - ratio) * SIZE) and te.num_rows == round(ratio * SIZE)
        comb = pd.concat([tr.to_pandas(), te.to_pandas()])
        assert df.sort_values("a").reset_index(drop=True).equals(comb.sort_values("a").reset_index(drop=True))
    h0, h0b, h1 = pbn.HoldOut(df, seed=0), pbn.HoldOut(df, seed=0), pbn.HoldOut(df, seed=1)
    assert h0.training_data().equals(h0b.training_data()) and h0.test_data().equals(h0b.test_data())
    assert not h0.training_data().equals(h1.training_data()) and not h0.test_data().equals(h1.test_data())
    dn = null_table(df)
    non_null = dn.dropna()
    hold = pbn.HoldOut(dn)
    tr, te = hold.training_data(), hold.test_data()
    assert tr.num_rows + te.num_rows == non_null.shape[0]
    assert tr.num_rows == round(0.8 * non_null.shape[0]) and te.num_rows == round(0.2 * non_null.shape[0])
    comb = pd.concat([tr.to_pandas(), te.to_pandas()])
    assert comb.sort_values("a").reset_index(drop=True).equals(non_null.sort_values("a").reset_index(drop=True))
    hn = pbn.HoldOut(dn, include_null=True)
    assert hn.training_data().num_rows == round(0.8 * SIZE) and hn.test_data().num_rows == round(0.2 * SIZE)
    for bad in (10, 0):                      # holdoutlikelihood_test.py:60-66 message
        with pytest.raises(ValueError, match="test_ratio must be a number"):
            pbn.HoldOut(df, bad)
