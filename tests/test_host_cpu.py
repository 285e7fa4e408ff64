"""CPU-side tests: the C-ABI library loads and exports every declared symbol,
fails loudly without a GPU, and the host logic (loader, writers, drop-in module
argument checks) matches the reference's behaviour recorded in the goldens."""
import json
import logging
import os
import re

import numpy as np
import pandas as p
import pytest

from desman_amd import _lib

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HAVE_GPU = _lib.device_count() > 0 if os.path.exists(_lib.LIB_PATH) else False


def test_library_exports_every_declared_symbol():
    hdr = open(_lib.HEADER_PATH).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b((?:dsm|c)_[A-Za-z0-9_]+)\s*\(", hdr))      # dsm_* and the reference-named c_* aliases
    declared -= {"dsm_ctx"}
    assert {"c_initRNG", "c_setRNG", "c_freeRNG", "c_sample_tau"} <= declared
    assert len(declared) >= 40
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), "libdesman_hip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "no ctypes signature for %s" % name
    assert set(_lib.SIGNATURES) <= declared
    assert lib.dsm_version().decode().startswith("desman_hip")
    assert lib.dsm_kernel_name(2).decode() == "tau_kernel"


@pytest.mark.skipif(HAVE_GPU, reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    with pytest.raises(_lib.DesmanHipError):
        _lib.Context(0)
    from desman_amd import sampletau
    with pytest.raises(_lib.DesmanHipError):
        sampletau.initRNG()
    tau = np.zeros((2, 2, 4), dtype=np.int64); tau[:, :, 0] = 1
    with pytest.raises(_lib.DesmanHipError):
        sampletau.sample_tau(tau, np.full((3, 2), 0.5), np.eye(4) * 0.96 + 0.01, np.ones((2, 3, 4), dtype=np.int64))


def test_sampletau_argument_checks_like_cython():
    from desman_amd import sampletau
    tau = np.zeros((2, 2, 4), dtype=np.int64); tau[:, :, 0] = 1
    pi = np.full((3, 2), 0.5); eta = np.eye(4) * 0.96 + 0.01
    var = np.ones((2, 3, 4), dtype=np.int64)
    with pytest.raises(TypeError):
        sampletau.sample_tau(None, pi, eta, var)
    with pytest.raises(TypeError):
        sampletau.sample_tau(tau.tolist(), pi, eta, var)
    with pytest.raises(ValueError):
        sampletau.sample_tau(tau.astype(np.int32), pi, eta, var)
    with pytest.raises(ValueError):
        sampletau.sample_tau(tau, pi.astype(np.float32), eta, var)
    with pytest.raises(ValueError):
        sampletau.sample_tau(tau[:, :, ::2], pi, eta, var)
    with pytest.raises(ValueError):
        sampletau.sample_tau(tau, pi, eta, var[:, :2, :].copy())      # shape cross-check (superset)
    with pytest.raises(TypeError):
        sampletau.setRNG("x")


def test_mt_seed_state_matches_numpy_legacy_seeding():
    for seed in (0, 1, 23724839):
        st = _lib.mt_seed_state(seed)
        bg = np.random.MT19937(); bg._legacy_seeding(seed if seed else 4357)
        assert np.array_equal(st[:624], bg.state["state"]["key"]) and st[624] == 624


def _frame(z):
    return p.DataFrame(z["frame_values"], index=z["frame_index"].tolist(), columns=z["frame_columns"].tolist())


def test_variant_filter_constructor_and_select_random():
    from desman_amd.Variant_Filter import Variant_Filter
    z = np.load(os.path.join(GOLDEN, "host_formats.npz"))
    flt = Variant_Filter(_frame(z), randomState=np.random.RandomState(238329), optimise=True, threshold=None,
                         min_coverage=5.0, qvalue_cutoff=1e-3)
    assert (flt.V, flt.S) == (int(z["V"]), int(z["S"]))
    assert np.array_equal(flt.sample_filter, z["sample_filter"])
    assert flt.sample_indices == z["sample_indices"].tolist()
    assert np.array_equal(flt.snps_filter, z["snps_filter"])
    assert np.array_equal(flt.eta, z["flt_eta"]) and np.array_equal(flt.selected, z["selected"])
    assert np.array_equal(np.asarray(flt.position), z["position"])
    flt.select_Random(5)
    assert np.array_equal(flt.snps_filter, z["sel_snps"]) and np.array_equal(flt.selected, z["sel_selected"])
    assert flt.selected_indices == z["sel_indices"].tolist()
    assert np.array_equal(flt.selected_indices_original, z["sel_indices_original"])
    if not HAVE_GPU:                                   # the -f filter runs on the GPU only: loud failure, no fallback
        with pytest.raises(_lib.DesmanHipError):
            flt.get_filtered_VariantsLogRatio()


def test_output_files_byte_identical_to_reference_writers(tmp_path):
    from desman_amd.Variant_Filter import Variant_Filter
    from desman_amd.Output_Results import Output_Results
    z = np.load(os.path.join(GOLDEN, "host_formats.npz"))
    want = json.loads(str(z["files"]))
    frame = _frame(z)
    flt = Variant_Filter(frame, randomState=np.random.RandomState(238329), optimise=True, threshold=None,
                         min_coverage=5.0, qvalue_cutoff=1e-3)
    flt.select_Random(5)

    class Stub:
        lp_star = -12345.678901

        def __init__(self, tau_star, pt):
            self.tau_star, self._pt = tau_star, pt
            self.V, self.G = tau_star.shape[0], tau_star.shape[1]

        def probabilisticTau(self):
            return self._pt

        def meanDeviance(self):
            return 24680.13579

    hs, hns = Stub(z["tau_star_s"], z["ptau_s"]), Stub(z["tau_star_ns"], z["ptau_ns"])
    d = str(tmp_path / "out")
    o = Output_Results(d)
    o.set_Variants(frame); o.set_Variant_Filter(flt); o.set_haplo_SNP(hs, 3)
    o.output_Filtered_Tau(hs.tau_star); o.output_Tau_Mean(z["ptau_s"])
    o.output_Gamma(z["gamma"]); o.output_Gamma_Mean(z["gamma"] * 0.5 + 0.25)
    o.output_Eta(z["eta"]); o.output_Eta_Mean(z["eta"].T.copy())
    o.output_Selected_Variants(); o.outPredFit(hns, 3); o.output_collated_Tau(hns, frame)
    logging.shutdown()
    got = {f: open(os.path.join(d, f)).read() for f in sorted(os.listdir(d)) if f != "log_file.txt"}
    assert sorted(got) == sorted(want)
    for f in want:
        assert got[f] == want[f], f


def test_fast_haplotype_table_text_is_pandas_text(tmp_path):
    """Output_Results._table_bytes (the vectorised writer of Filtered_Tau_star / Tau_Mean / Collated_*) against DataFrame.to_csv on
    values that are awkward to print, and its refusals (pandas then writes the file)"""
    import pandas as pd
    from desman_amd.Output_Results import _table_bytes
    rng = np.random.default_rng(3)
    V, G = 257, 5
    names = ["contig_%d" % (i // 7) for i in range(V)]
    pos = rng.integers(0, 10 ** 7, size=V)
    pool = np.array([0.0, 1.0, 0.1 + 0.2, 1e-05, 1e-300, 123456789.125, 2.0 / 3.0, 5e-324, 1e16, 0.002, 1.0 / 500.0 * 499.0, -0.0, -1.5])
    cases = [pool[rng.integers(0, pool.size, size=(V, G * 4))], rng.integers(0, 2, size=(V, G * 4)), rng.integers(-5, 10 ** 12, size=(V, 3)),
             (rng.integers(0, 501, size=(V, G * 4)) / 500.0)]
    for flat in cases:
        frame = pd.DataFrame(flat, index=names)
        frame['Position'] = pos
        order = frame.columns.tolist()
        want = frame[order[-1:] + order[:-1]].to_csv().encode()
        assert _table_bytes(flat, names, pos) == want
    ok = cases[1]
    assert _table_bytes(ok, ["a,b"] + names[1:], pos) is None                 # a name that needs quoting
    assert _table_bytes(ok, ["contig_\u00e9"] + names[1:], pos) is None        # a non-ASCII name (raised UnicodeEncodeError: ADVICE r4)
    assert _table_bytes(ok, names, pos.astype(np.float64)) is None            # positions that are not integers
    assert _table_bytes(np.where(ok == 0, np.nan, 1.0), names, pos) is None   # missing values
    assert _table_bytes(rng.random((V, 300)), names, pos) is None             # too many distinct values to pay


def test_cli_flags_and_quirks():
    from desman_amd.cli import build_parser
    a = build_parser().parse_args(["x.freq", "-g", "4"])
    assert (a.no_iter, a.random_select, a.filter_variants, a.random_seed, a.min_coverage) == \
        (None, None, None, 23724839, 5.0)                   # -i absent -> None -> sampler default 250
    a = build_parser().parse_args(["x.freq", "-g", "4", "-i", "-r", "-f", "-v", "-p", "False"])
    assert (a.no_iter, a.random_select, a.filter_variants, a.min_variant_freq) == (250, 1000, 3.84, 0.01)
    assert a.optimiseP is True                              # type=bool quirk: any string is True


# ---------------------------------------------------------------- f4 host logic (no GPU)
@pytest.mark.parametrize("name", ["gene_assign", "gene_assign_lowcov"])
def test_compgenes_matches_reference(name):
    """GeneAssign.compGenes (greedy genome matching) against the reference function's output."""
    import os
    from desman_amd.GeneAssign import compGenes
    from desman_amd.synth import synth_genes
    import ast
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    d = synth_genes(int(z['C']), int(z['S']), int(z['G']), seed=int(z['synth_seed']), **dict(ast.literal_eval(str(z['synth_kw']))))
    tot, acc, full = compGenes(np.rint(z['kl_eta']), d['eta_true'].astype(np.float64))
    assert tot == float(z['comp_total'])
    np.testing.assert_array_equal(acc, z['comp_acc'])
    np.testing.assert_array_equal(full, z['comp_matrix'])


def test_geneassign_parser_has_reference_flags_and_defaults():
    from desman_amd.GeneAssign import build_parser, expand_sample_names
    a = build_parser().parse_args(["scg.csv", "gamma.csv", "cov.csv", "eps.csv"])
    assert (a.random_seed, a.eta_max, a.iter_max, a.var_max, a.output_stub) == (23724839, 2, 20, 1e10, "output")
    assert a.genomes is None and a.variant_file is None and a.assign_tau is False and a.rng == "mt19937"
    b = build_parser().parse_args(["a", "b", "c", "d", "-s", "5", "-e", "3", "-i", "7", "-m", "40", "-o", "x", "-g", "g.csv",
                                   "-v", "v.csv", "--assign_tau"])
    assert (b.random_seed, b.eta_max, b.iter_max, b.var_max, b.output_stub, b.genomes, b.variant_file, b.assign_tau) == \
        (5, 3, 7, 40, "x", "g.csv", "v.csv", True)
    assert expand_sample_names(["s1", "s2"]) == ["s1-A", "s1-C", "s1-G", "s1-T", "s2-A", "s2-C", "s2-G", "s2-T"]


def test_gene_sampler_fails_loudly_without_a_gpu():
    """no CPU fallback on the gene path either: without a device the constructor raises DesmanHipError"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import pandas as pd
    from desman_amd import _lib
    from desman_amd.Eta_Sampler import Eta_Sampler
    from desman_amd.GeneAssign import KLAssign
    cov = pd.DataFrame(np.ones((3, 2)), index=["a", "b", "c"], columns=["s0", "s1"])
    with pytest.raises(_lib.DesmanHipError):
        Eta_Sampler(np.random.RandomState(1), None, cov, np.full((2, 2), 0.5), np.ones((2, 2)), np.ones(2),
                    np.eye(4), np.ones((3, 2)))
    with pytest.raises(_lib.DesmanHipError):
        KLAssign(np.random.RandomState(1), np.ones((3, 2)), np.ones((2, 2))).factorize()


def test_cli_rejects_assign_file_before_any_work(tmp_path):
    """-a is dead upstream (bin/desman:213-214 stops in ipdb after the whole run): rejected right after parsing,
    before the output directory or any GPU work"""
    from desman_amd import cli
    out = tmp_path / "never_created"
    with pytest.raises(SystemExit) as e:
        cli.main([str(tmp_path / "missing.freq"), "-g", "3", "-o", str(out), "-a", str(tmp_path / "x.csv")])
    assert "assign_file" in str(e.value)
    assert not out.exists()


def test_vshard_bounds_cover_the_table_once():
    from desman_amd import vshard
    for v_total, world in ((10, 3), (50000, 8), (7, 8), (1, 1), (12289, 4)):
        b = vshard.shard_bounds(v_total, world)
        assert b[0] == 0 and b[-1] == v_total and len(b) == world + 1 and all(x <= y for x, y in zip(b, b[1:]))
        sizes = [y - x for x, y in zip(b, b[1:])]
        assert max(sizes) - min(sizes) <= 1
    v = vshard._DevView(4096, 18, "<f8").__cuda_array_interface__
    assert v["shape"] == (18,) and v["data"] == (4096, False) and v["version"] == 2
