"""Row f3: the likelihood-ratio variant filter (Variant_Filter.py:320-390).
CPU: the oracle's bounded-Brent restatement against the installed SciPy, and the host logic
(outer loop, BH q-values, outputs) driven by the oracle step, against goldens from the reference.
GPU: the same with the real lrt_kernel."""
import os

import numpy as np
import pandas as p
import pytest
from scipy.optimize import minimize_scalar

from desman_amd import _lib
from desman_amd import Variant_Filter as vfm
from oracle import ref_numpy as rn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _frame(z):
    return p.DataFrame(z["frame_values"], index=z["frame_index"].tolist(), columns=z["frame_columns"].tolist())


def test_oracle_fminbound_equals_scipy_bounded():
    rng = np.random.default_rng(0)
    eta = rng.dirichlet(np.ones(4), size=4) * 0.1 + 0.9 * np.eye(4)
    for _ in range(400):
        f = rng.multinomial(rng.integers(5, 5000), rng.dirichlet([5, 1, .1, .1])).astype(float)
        n, m = np.argsort(-f)[:2]
        fn = lambda x: rn.mix_nll(x, eta, n, m, f)          # noqa: E731
        assert rn.fminbound(fn, 0.0, 0.99) == minimize_scalar(fn, bounds=(0.0, 0.99), method='bounded').x


def test_benjamini_hochberg_matches_reference_procedure():
    z = np.load(os.path.join(GOLDEN, "variant_filter_lrt.npz"))
    for tag in ("opt", "noopt"):
        np.testing.assert_allclose(vfm.benjamini_Hochberg(z[tag + "_pvalue"]), z[tag + "_qvalue"], rtol=1e-13, atol=0)
    q = vfm.benjamini_Hochberg(np.array([0.01, 0.01, 0.5, 0.0, 1.0]))
    assert q.shape == (5,) and (q >= np.array([0.01, 0.01, 0.5, 0.0, 1.0]) - 1e-15).all()


def _run(z, tag, monkeypatch, use_oracle):
    if use_oracle:
        monkeypatch.setattr(_lib, "lrt_step", lambda ffreq, a, b, eta, up, opt, pv, device=0:
                            rn.lrt_step(np.asarray(ffreq, dtype=np.float64), np.asarray(a), np.asarray(b), eta, up, opt, pv))
    frame = _frame(z)
    f = vfm.Variant_Filter(frame, randomState=np.random.RandomState(1), optimise=(tag == "opt"), threshold=3.84,
                           min_coverage=5.0, qvalue_cutoff=1.0e-3)
    f.get_filtered_VariantsLogRatio()
    # fp tolerance: identical algorithm, different log implementations / summation order
    np.testing.assert_allclose(f.ratioNLL, z[tag + "_ratio"], rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(f.pvalue, z[tag + "_pvalue"], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(f.qvalue, z[tag + "_qvalue"], rtol=1e-7, atol=1e-12)
    assert np.array_equal(f.filtered, z[tag + "_filtered"]) and np.array_equal(f.selected, z[tag + "_selected"])
    np.testing.assert_allclose(f.eta, z[tag + "_eta"], rtol=1e-12)
    np.testing.assert_allclose(f.minV, z[tag + "_minV"], rtol=1e-14)
    assert np.array_equal(f.snps_filter, z[tag + "_snps"])
    np.testing.assert_allclose(f.calc_Error_Matrix(), z[tag + "_tran"], rtol=1e-12)
    assert f.selected_variants_todf(frame).to_csv() == str(z[tag + "_selvar_csv"])


@pytest.mark.parametrize("tag", ["opt", "noopt"])
def test_filter_host_logic_with_oracle_step(tag, monkeypatch):
    _run(np.load(os.path.join(GOLDEN, "variant_filter_lrt.npz")), tag, monkeypatch, use_oracle=True)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["opt", "noopt"])
def test_filter_gpu_matches_reference(tag, monkeypatch):
    _run(np.load(os.path.join(GOLDEN, "variant_filter_lrt.npz")), tag, monkeypatch, use_oracle=False)


@pytest.mark.gpu
def test_lrt_kernel_vs_oracle_and_cog_counts():
    z = np.load(os.path.join(GOLDEN, "variant_filter_lrt.npz"))
    freq = z["cog_freq"]
    V = freq.shape[0]
    maxA = np.argmax(freq, axis=1)
    ft = freq.copy(); ft[np.arange(V), maxA] = -1
    maxB = np.argmax(ft, axis=1)
    eta = 0.96 * np.eye(4) + 0.01
    ff = freq.astype(np.float64)
    p0 = np.minimum(freq.max(axis=1) / np.maximum(freq.sum(axis=1), 1), 0.99)
    pg, mg, bg = _lib.lrt_step(ff, maxA, maxB, eta, 0.99, True, p0)
    sub = np.arange(0, V, 7)                                    # the python oracle is slow: every 7th position
    po, mo, bo = rn.lrt_step(ff[sub], maxA[sub], maxB[sub], eta, 0.99, True, p0[sub])
    np.testing.assert_allclose(pg[sub], po, rtol=0, atol=1e-9)   # same Brent path; only log rounding differs
    np.testing.assert_allclose(mg[sub], mo, rtol=1e-12)
    np.testing.assert_allclose(bg[sub], bo, rtol=1e-12)
    # the whole filter on the COG0015 base-count sums selects the reference's 27 positions
    class F(vfm.Variant_Filter):
        def __init__(self):
            self.freq = freq; self.ffreq = ff; self.V = V; self.S = 1
            self.snps_filter = freq[:, None, :]
            self.threshold, self.qvalue_cutoff, self.optimise = 3.84, 1.0e-3, True
            self.max_iter, self.Nthreshold, self.upperP, self.device = 100, 10, 0.99, 0
            self.eta = eta.copy()
    f = F()
    f.get_filtered_VariantsLogRatio()
    assert f.NS == int(z["cog_nsel"]) == 27
    assert np.array_equal(f.filtered, z["cog_filtered"])
    np.testing.assert_allclose(f.eta, z["cog_eta"], rtol=1e-12)
    np.testing.assert_allclose(f.ratioNLL, z["cog_ratio"], rtol=1e-9, atol=1e-6)


@pytest.mark.gpu
def test_cli_with_filter_flag(tmp_path):
    from desman_amd.cli import main
    import sys
    sys.path.insert(0, GOLDEN)
    z = np.load(os.path.join(GOLDEN, "variant_filter_lrt.npz"))
    frame = _frame(z)
    frame.index.name = "Contig"
    freq = str(tmp_path / "lrt.freq")
    frame.to_csv(freq)
    out = str(tmp_path / "o")
    main([freq, "-g", "2", "-i", "10", "-f", "-p", "1", "-o", out])
    sel = p.read_csv(os.path.join(out, "Selected_variants.csv"), index_col=0)
    assert sel.shape[0] == int(z["opt_selected"].sum())
    ts = p.read_csv(os.path.join(out, "Filtered_Tau_star.csv"), index_col=0)
    assert ts.shape[0] == sel.shape[0]
    # the stand-alone tool writes the reference's file set
    stub = str(tmp_path / "vf_")
    vfm.main([freq, "-o", stub, "-p", "-f", "3.84"])
    for name in ("sel_var.csv", "v_df.csv", "p_df.csv", "q_df.csv", "r_df.csv", "tran_df.csv"):
        assert os.path.exists(stub + name)
    assert open(stub + "sel_var.csv").read() == str(z["opt_selvar_csv"])
