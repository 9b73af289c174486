""""Next" rows of SURVEY 8(f): `save` (byte-for-byte against the reference's fixtures test/data/save1-4,
test/runtests.jl:185-203) and `split` (group sizes / fold rule; the reference's own test is skipped)."""
import os

import numpy as np

import simspread_jl_amd as ss

GOLD = os.path.join(os.path.dirname(__file__), "golden", "save")


def test_save_matches_reference_fixtures(tmp_path):
    y = ss.NamedMatrix(np.array([[1, 0, 1], [0, 1, 0]]), ["s1", "s2"], ["t1", "t2", "t3"])
    yhat = y.copy(); yhat.integer = True
    cases = [("save1", (y, yhat), {}), ("save2", (y, yhat), {"delimiter": " "}),
             ("save3", (1, y, yhat), {}), ("save4", (1, y, yhat), {"delimiter": " "})]
    for name, args, kw in cases:
        p = tmp_path / name
        ss.save(str(p), *args, **kw)
        with open(os.path.join(GOLD, name)) as f:
            assert p.read_text().splitlines() == f.read().splitlines(), name
    # append mode, float payloads print like Julia (0.5, -99.0)
    p = tmp_path / "float"
    yh = ss.NamedMatrix([[0.0, 0.5, -99.0]], ["q1"], ["t1", "t2", "t3"])
    yy = ss.NamedMatrix([[0.0, 1.0, 0.0]], ["q1"], ["t1", "t2", "t3"])
    ss.save(str(p), 3, yh, yy); ss.save(str(p), 4, yh, yy)
    assert p.read_text().splitlines()[1] == '3\t"q1"\t"t2"\t0.5\t1.0'
    assert p.read_text().splitlines()[5] == '4\t"q1"\t"t3"\t-99.0\t0.0'


def test_split_fold_rule():
    y = ss.NamedMatrix(np.zeros((10, 5)), [f"s{i}" for i in range(1, 11)], [f"t{i}" for i in range(1, 6)])
    groups = ss.split(y, 5, seed=1)
    assert [len(g) for g in groups] == [2] * 5
    assert sorted(n for g in groups for n in g) == sorted(y.names(1))
    assert ss.split(y, 5, seed=1) == groups and ss.split(y, 5, seed=2) != groups
    g3 = ss.split(y, 3)
    assert sorted(len(g) for g in g3) == [3, 3, 4]   # source i -> fold mod(i, k) + 1
