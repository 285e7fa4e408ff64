"""Row f2: desman_amd.resolvenhap against the reference's scripts/resolvenhap.py
(stdout line and every *R.csv it writes), on synthetic sweep trees (CPU only)."""
import glob
import io
import json
import os
import contextlib

import numpy as np
import pytest

from desman_amd import resolvenhap

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _build(root, spec):
    # the tree builder is shared with the fixture generator
    import importlib.util
    sp = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    src = open(os.path.join(GOLDEN, "make_golden.py")).read()
    ns = {"os": os, "np": np}
    start = src.index("def build_sweep_tree")
    end = src.index("def gen_resolvenhap")
    exec(src[start:end], ns)
    return ns["build_sweep_tree"](root, spec)


@pytest.mark.parametrize("case", [0, 1, 2])
def test_resolvenhap_matches_reference_script(tmp_path, case):
    z = np.load(os.path.join(GOLDEN, "resolvenhap_%d.npz" % case))
    spec = {k: z[k] for k in z.files if k not in ("stdout", "rfiles")}
    stub = _build(str(tmp_path), spec)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = resolvenhap.resolve(stub)
    assert buf.getvalue().replace(str(tmp_path), "<ROOT>") == str(z["stdout"])
    want = json.loads(str(z["rfiles"]))
    got = {os.path.relpath(f, str(tmp_path)): open(f).read() for f in glob.glob(os.path.join(str(tmp_path), "*", "*R.csv"))}
    assert sorted(got) == sorted(want)
    for f in want:
        assert got[f] == want[f], f
    assert res[0] == int(str(z["stdout"]).split(",")[0])


def test_comp_snd():
    rng = np.random.default_rng(0)
    i1, i2 = rng.integers(0, 4, (50, 3)), rng.integers(0, 4, (50, 4))
    t1 = np.eye(4, dtype=int)[i1]; t2 = np.eye(4, dtype=int)[i2]
    d = resolvenhap.comp_snd(t1, t2)
    for g in range(3):
        for h in range(4):
            assert d[g, h] == (i1[:, g] != i2[:, h]).sum()
