"""Oracle of the batch collation against the vectors recorded from the unmodified reference."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import collate_recipe as R
from oracle import collate_oracle as O

GOLD = np.load(os.path.join(HERE, "golden", "collate.npz"))
ELEM_ID = {el: k for k, el in enumerate(R.ELEMENTS)}


@pytest.mark.parametrize("case", R.CASES, ids=[c[0] for c in R.CASES])
def test_collate_oracle_matches_reference(case):
    name, n_graphs, seed, fmt, comps_as, max_nbr, target, batches = case
    data = R.make_dataset(n_graphs, seed, fmt, comps_as)
    table = R.embedding_table()
    for bi, ids in enumerate(batches):
        got = O.collate(data, ids, table, ELEM_ID, max_nbr, target)
        for k, v in got.items():
            ref = GOLD[f"{name}.b{bi}.{k}"]
            assert v.shape == ref.shape, (k, v.shape, ref.shape)
            assert v.dtype == ref.dtype, (k, v.dtype, ref.dtype)
            assert np.array_equal(v, ref), k            # bit-exact: index work and exact fp32 gathers / divisions
