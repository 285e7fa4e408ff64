#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by IMPORTING the reference.

Runs only in the development container, where /root/reference exists; the
fixtures it writes (inputs + expected outputs, small .npz files) are what
travels.  No reference source is copied.

Import shims (environment only, nothing of the reference is modified):
  * numpy >= 1.24 dropped ``np.int`` / ``np.float`` that the reference uses
    (e.g. HaploSNP_Sampler.py:68, Init_NMFT.py:49) -> aliased to int / float.
  * desman/__init__.py calls pkg_resources.require("desman") (needs an
    installed dist) -> a stub package object pointing at the source dir.
  * desman/HaploSNP_Sampler.py does ``import sampletau`` (the Cython/GSL
    extension, which cannot be built here: no GSL).  A module object named
    ``sampletau`` backed by oracle/ (our C restatement) is registered so the
    import succeeds; fixtures that depend on it say so in their ``note``.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--cog]
"""
import argparse
import glob
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

np.int = int      # noqa
np.float = float  # noqa

from oracle import cbind  # noqa: E402

REF = "/root/reference"


def import_reference():
    pkg = types.ModuleType("desman")
    pkg.__path__ = [os.path.join(REF, "desman")]
    sys.modules["desman"] = pkg
    st = types.ModuleType("sampletau")
    st.initRNG = cbind.initRNG
    st.setRNG = cbind.setRNG
    st.freeRNG = cbind.freeRNG
    st.sample_tau = cbind.sample_tau
    sys.modules["sampletau"] = st
    import warnings
    warnings.simplefilter("ignore")
    import desman.Init_NMFT as inmft
    import desman.HaploSNP_Sampler as hsnp
    import desman.Desman_Utils as du
    return inmft, hsnp, du


def synth(V, S, G, seed):
    from desman_amd.synth import synth_counts
    return synth_counts(V, S, G, seed)[0]


def make_sampler(hsnp, counts, G, seed, max_iter=3):
    rs = np.random.RandomState(seed)
    return hsnp.HaploSNP_Sampler(counts, G, rs, max_iter=max_iter)


def gen_tau_sweep(hsnp, out):
    """Conditional log-probs + draws of the tau sweep from the reference's own
    pure-Python sampler (HaploSNP_Sampler.sampleTau, :148-183).  Its categorical
    draw (sampleLogProb, :124-127) is replaced by the native sweep's inverse-CDF
    rule (c_sample_tau.c:48-91,172-176) fed with recorded uniforms, so the
    fixture pins the conditional of the C sweep: logp[V,G,4], tau_out."""
    for (V, S, G, seed) in [(32, 8, 3, 11), (64, 16, 5, 12), (40, 64, 8, 13), (33, 7, 3, 14)]:
        counts = synth(V, S, G, 100 + seed)
        smp = make_sampler(hsnp, counts, G, seed)
        rng = np.random.default_rng(seed)
        smp.gamma = np.ascontiguousarray(rng.dirichlet(np.ones(G), size=S))
        smp.eta = np.ascontiguousarray(rng.dirichlet(np.ones(4), size=4) * 0.08 + 0.92 * np.eye(4))
        smp.updateTauIndices()
        tau_in = smp.tau.copy()
        u = cbind.MT19937(1000 + seed).uniform(V * G)
        rec = []

        def pick(logp, _u=iter(u), _rec=rec):
            _rec.append(np.array(logp, dtype=np.float64))
            p = np.exp(logp - np.max(logp))
            p = p / p.sum()
            c = np.cumsum(p)
            x = next(_u)
            return 0 if x < c[0] else 1 if x < c[1] else 2 if x < c[2] else 3

        smp.sampleLogProb = pick
        smp.sampleTau()
        np.savez_compressed(os.path.join(out, "tau_sweep_V%d_S%d_G%d.npz" % (V, S, G)),
                            counts=counts, tau_in=tau_in, gamma=smp.gamma, eta=smp.eta, u=u,
                            mt_seed=1000 + seed, logp=np.array(rec).reshape(V, G, 4),
                            tau_out=smp.tau.copy(),
                            note="reference python sampleTau conditionals; inverse-CDF draw")


def gen_loglik(hsnp, out):
    rows = []
    for (V, S, G, seed) in [(20, 6, 2, 21), (30, 16, 5, 22), (16, 64, 8, 23)]:
        counts = synth(V, S, G, 200 + seed)
        smp = make_sampler(hsnp, counts, G, seed)
        rng = np.random.default_rng(seed)
        gamma = np.ascontiguousarray(rng.dirichlet(np.ones(G), size=S))
        eta = np.ascontiguousarray(rng.dirichlet(np.ones(4), size=4) * 0.08 + 0.92 * np.eye(4))
        ll = smp.logLikelihood(gamma, smp.tau, eta)
        lp = smp.logPosterior(gamma, smp.tau, eta)
        rows.append(dict(counts=counts, tau=smp.tau.copy(), gamma=gamma, eta=eta, ll=ll, lp=lp))
    np.savez_compressed(os.path.join(out, "loglik.npz"),
                        **{"%s_%d" % (k, i): r[k] for i, r in enumerate(rows) for k in r}, n=len(rows))


def gen_degenerate(hsnp, out):
    rows = []
    for case, (V, S, G, dup) in enumerate([(25, 5, 4, [(0, 2)]), (25, 5, 5, [(1, 3), (1, 4)]),
                                           (25, 5, 3, []), (10, 4, 6, [(0, 1), (2, 3), (2, 5)])]):
        counts = synth(V, S, G, 300 + case)
        smp = make_sampler(hsnp, counts, G, 30 + case)
        for (g, h) in dup:
            smp.tau[:, h, :] = smp.tau[:, g, :]
        smp.updateTauIndices()
        tau_in, gamma_in = smp.tau.copy(), smp.gamma.copy()
        smp.removeDegenerate()
        rows.append(dict(tau_in=tau_in, gamma_in=gamma_in, tau_out=smp.tau.copy(),
                         gamma_out=smp.gamma.copy(), G_out=smp.G))
    np.savez_compressed(os.path.join(out, "degenerate.npz"),
                        **{"%s_%d" % (k, i): r[k] for i, r in enumerate(rows) for k in r}, n=len(rows))


def gen_gibbs_pieces(hsnp, out):
    """sampleMu / sampleGamma / sampleEta with the reference's RandomState stream
    (pins oracle/ref_numpy.py), plus a short update() trajectory.  The
    trajectory's tau sweep comes from oracle/ (registered as ``sampletau``)."""
    V, S, G, seed = 12, 5, 3, 41
    counts = synth(V, S, G, 400)
    smp = make_sampler(hsnp, counts, G, seed, max_iter=4)
    tau0, gamma0, eta0 = smp.tau.copy(), smp.gamma.copy(), smp.eta.copy()
    state0 = smp.randomState.get_state()
    smp.sampleMu(smp.tau, smp.gamma, smp.eta)
    E1, mu1 = smp.E.copy(), smp.mu.copy()
    smp.sampleGamma()
    gamma1 = smp.gamma.copy()
    smp.sampleEta()
    eta1 = smp.eta.copy()
    # trajectory from a fresh, identically seeded sampler
    smp2 = make_sampler(hsnp, counts, G, seed, max_iter=4)
    cbind.initRNG(); cbind.setRNG(seed)
    smp2.update()
    np.savez_compressed(
        os.path.join(out, "gibbs_pieces.npz"), counts=counts, G=G, seed=seed, tau0=tau0,
        gamma0=gamma0, eta0=eta0, rs_key=state0[1], rs_pos=state0[2], E1=E1, mu1=mu1,
        gamma1=gamma1, eta1=eta1, ll_store=smp2.ll_store.copy(), gamma_store=smp2.gamma_store.copy(),
        eta_store=smp2.eta_store.copy(), tau_final=smp2.tau.copy(), tau_star=smp2.tau_star.copy(),
        gamma_star=smp2.gamma_star.copy(), eta_star=smp2.eta_star.copy(), lp_star=smp2.lp_star,
        tau_mean=smp2.tauMean(), mean_dev=smp2.meanDeviance(),
        note="update() trajectory: python parts = reference, tau sweep = oracle C restatement")


def gen_nmft(inmft, out):
    for (V, S, G, seed) in [(24, 6, 3, 51), (50, 16, 5, 52), (30, 64, 8, 53), (20, 8, 1, 54)]:
        counts = synth(V, S, max(G, 2), 500 + seed)
        rs = np.random.RandomState(seed)
        nm = inmft.Init_NMFT(counts, G, rs)
        F = nm.freq_matrix.copy()
        nm.random_initialize()
        tau_raw, gamma_raw = nm.tau.copy(), nm.gamma.copy()
        nm._adjustment()
        snaps = {}
        div0 = nm.div_objective()
        for it in range(1, 101):
            nm.div_update()
            nm._adjustment()
            if it in (1, 10, 100):
                snaps["tau_%d" % it] = nm.tau.copy()
                snaps["gamma_%d" % it] = nm.gamma.copy()
                snaps["div_%d" % it] = nm.div_objective()
        get_tau = nm.get_tau()
        # full factorize() with a small iteration cap, from the same seed
        rs2 = np.random.RandomState(seed)
        nm2 = inmft.Init_NMFT(counts, G, rs2, max_iter=300)
        nm2.factorize()
        # factorize_tau with gamma fixed
        rs3 = np.random.RandomState(seed + 1000)
        nm3 = inmft.Init_NMFT(counts, G, rs3, max_iter=50)
        nm3.gamma = nm2.gamma.copy()
        nm3.random_initialize_tau()
        tau3_raw = nm3.tau.copy()
        rs3b = np.random.RandomState(seed + 1000)
        nm3b = inmft.Init_NMFT(counts, G, rs3b, max_iter=50)
        nm3b.gamma = nm2.gamma.copy()
        nm3b.factorize_tau()
        np.savez_compressed(
            os.path.join(out, "nmft_V%d_S%d_G%d.npz" % (V, S, G)), counts=counts, G=G, seed=seed,
            F=F, tau_raw=tau_raw, gamma_raw=gamma_raw, div0=div0, get_tau_100=get_tau,
            fact_tau=nm2.tau.copy(), fact_gamma=nm2.gamma.copy(), fact_get_tau=nm2.get_tau(),
            fact_div=nm2.div_objective(), ft_tau_raw=tau3_raw, ft_tau=nm3b.tau.copy(),
            ft_get_tau=nm3b.get_tau(), **snaps)


def small_freq_frame():
    """a tiny .freq-style frame: 9 positions, 4 samples (one below the coverage cut)."""
    import pandas as p
    rng = np.random.default_rng(77)
    V, S = 9, 4
    cols = ["Position"]
    for s in range(S):
        cols += ["S%d-%s" % (s, b) for b in "ACGT"]
    data = np.zeros((V, 1 + 4 * S), dtype=np.int64)
    data[:, 0] = np.arange(V) * 13 + 5
    for s in range(S):
        depth = 3 if s == 2 else 40 + 10 * s
        data[:, 1 + 4 * s:5 + 4 * s] = rng.multinomial(depth, [0.6, 0.25, 0.1, 0.05], size=V)
    idx = ["gene%d" % (v // 3) for v in range(V)]
    return p.DataFrame(data, index=idx, columns=cols)


def gen_host_formats(out):
    """Variant_Filter constructor / select_Random and every Output_Results file,
    produced by the reference classes on a tiny frame and a stub sampler."""
    import json
    import tempfile
    import logging
    import desman.Variant_Filter as vf
    import desman.Output_Results as outr
    frame = small_freq_frame()
    flt = vf.Variant_Filter(frame, randomState=np.random.RandomState(238329), optimise=True, threshold=None,
                            min_coverage=5.0, qvalue_cutoff=1e-3)
    rec = dict(sample_filter=flt.sample_filter, sample_indices=np.array(flt.sample_indices),
               snps_filter=np.ascontiguousarray(flt.snps_filter), flt_eta=flt.eta, V=flt.V, S=flt.S,
               selected=flt.selected.copy(), position=np.asarray(flt.position))
    flt.select_Random(5)
    rec.update(sel_snps=np.ascontiguousarray(flt.snps_filter), sel_selected=flt.selected.copy(),
               sel_indices=np.array(flt.selected_indices), sel_indices_original=np.array(flt.selected_indices_original))

    class Stub:
        pass
    rng = np.random.default_rng(3)
    G = 2
    def stub(V):
        h = Stub()
        h.V, h.G = V, G
        idx = rng.integers(0, 4, size=(V, G))
        h.tau_star = np.zeros((V, G, 4), dtype=np.int64)
        np.put_along_axis(h.tau_star, idx[..., None], 1, axis=2)
        pt = rng.dirichlet(np.ones(4), size=(V, G))
        h.probabilisticTau = lambda pt=pt: pt
        h.lp_star = -12345.678901
        h.meanDeviance = lambda: 24680.13579
        return h, pt
    hs, pt = stub(5)
    hns, ptn = stub(flt.V - 5)
    gamma = rng.dirichlet(np.ones(G), size=flt.S)
    eta = rng.dirichlet(np.ones(4), size=4)
    with tempfile.TemporaryDirectory() as d:
        o = outr.Output_Results(d)
        o.set_Variants(frame); o.set_Variant_Filter(flt); o.set_haplo_SNP(hs, 3)
        o.output_Filtered_Tau(hs.tau_star); o.output_Tau_Mean(pt)
        o.output_Gamma(gamma); o.output_Gamma_Mean(gamma * 0.5 + 0.25)
        o.output_Eta(eta); o.output_Eta_Mean(eta.T.copy())
        o.output_Selected_Variants(); o.outPredFit(hns, 3); o.output_collated_Tau(hns, frame)
        logging.shutdown()
        files = {f: open(os.path.join(d, f)).read() for f in sorted(os.listdir(d)) if f != "log_file.txt"}
    np.savez_compressed(os.path.join(out, "host_formats.npz"), frame_values=frame.to_numpy(),
                        frame_index=np.array(frame.index.tolist()), frame_columns=np.array(frame.columns.tolist()),
                        tau_star_s=hs.tau_star, ptau_s=pt, tau_star_ns=hns.tau_star, ptau_ns=ptn, gamma=gamma,
                        eta=eta, files=json.dumps(files), **rec)


def build_sweep_tree(root, spec):
    """materialise a `<stub>_<G>_<r>/` tree from arrays (shared by the generator and the test)."""
    import pandas as p
    stub = os.path.join(root, "sw")
    V, S = int(spec["V"]), int(spec["S"])
    pos = np.arange(V) * 3 + 1
    idx = ["c%d" % (v // 7) for v in range(V)]
    for key in [k for k in spec if k.startswith("fit_")]:
        _, G, r = key.split("_")
        G, r = int(G), int(r)
        d = "%s_%d_%d" % (stub, G, r)
        os.makedirs(d, exist_ok=True)
        gt, ht, ll, pd = spec[key]
        with open(os.path.join(d, "fit.txt"), "w") as f:
            f.write("Fit,%d,%d,%f,%f\n" % (int(gt), int(ht), ll, pd))
        tau = spec["tau_%d_%d" % (G, r)]
        Gh = tau.shape[1]
        for name, arr in (("Filtered_Tau_star", tau.reshape(V, Gh * 4)),
                          ("Tau_Mean", spec["taum_%d_%d" % (G, r)].reshape(V, Gh * 4))):
            df = p.DataFrame(arr, index=idx)
            df['Position'] = pos
            c = df.columns.tolist()
            df[c[-1:] + c[:-1]].to_csv(os.path.join(d, name + ".csv"))
        for name in ("Gamma_star", "Gamma_mean"):
            p.DataFrame(spec["gam_%d_%d" % (G, r)], index=["S%d" % s for s in range(S)]).to_csv(os.path.join(d, name + ".csv"))
    return stub


def gen_resolvenhap(out):
    """scripts/resolvenhap.py run (unmodified, via runpy) on synthetic sweep trees: stdout + the *R.csv it writes."""
    import io
    import json
    import runpy
    import tempfile
    import contextlib
    rng = np.random.default_rng(11)
    cases = {}
    for case, (gvals, reps, pdfun, drop) in enumerate([
            (range(2, 7), 3, lambda G: 1000.0 * (0.5 ** min(G, 4)) + 3.0 * G, {(5, 1)}),
            (range(2, 4), 2, lambda G: 500.0 / G, set()),
            (range(1, 8), 4, lambda G: 900.0 - 100.0 * G, {(3, 0), (6, 2)})]):
        V, S = 60, 6
        spec = dict(V=V, S=S)
        base = rng.integers(0, 4, size=(V, 8))
        for G in gvals:
            for r in range(reps):
                ht = G - 1 if (G, r) in drop else G
                idx = base[:, rng.permutation(G)[:ht]].copy()
                flip = rng.random(idx.shape) < (0.02 + 0.08 * (G >= 5))
                idx[flip] = rng.integers(0, 4, size=int(flip.sum()))
                tau = np.zeros((V, ht, 4), dtype=np.int64)
                np.put_along_axis(tau, idx[..., None], 1, axis=2)
                spec["tau_%d_%d" % (G, r)] = tau
                spec["taum_%d_%d" % (G, r)] = rng.dirichlet(np.ones(4), size=(V, ht))
                spec["gam_%d_%d" % (G, r)] = rng.dirichlet(np.ones(ht) * 0.7, size=S)
                spec["fit_%d_%d" % (G, r)] = np.array([G, ht, -5000.0 - G, pdfun(G) + rng.normal(0, 2.0)])
        with tempfile.TemporaryDirectory() as d:
            stub = build_sweep_tree(d, spec)
            buf = io.StringIO()
            argv = sys.argv
            sys.argv = ["resolvenhap.py", stub]
            try:
                with contextlib.redirect_stdout(buf):
                    runpy.run_path(os.path.join(REF, "scripts", "resolvenhap.py"), run_name="__main__")
            finally:
                sys.argv = argv
            line = buf.getvalue().replace(d, "<ROOT>")
            rfiles = {}
            for f in sorted(glob.glob(os.path.join(d, "*", "*R.csv"))):
                rfiles[os.path.relpath(f, d)] = open(f).read()
        spec["stdout"] = line
        spec["rfiles"] = json.dumps(rfiles)
        cases[case] = spec
        np.savez_compressed(os.path.join(out, "resolvenhap_%d.npz" % case), **spec)


def lrt_frame(V=260, S=6, seed=5):
    """a .freq-style frame with ~15 % two-allele positions (the rest: sequencing errors only)."""
    import pandas as p
    rng = np.random.default_rng(seed)
    data = np.zeros((V, 1 + 4 * S), dtype=np.int64)
    data[:, 0] = np.arange(V) * 11 + 2
    for v in range(V):
        major = rng.integers(0, 4)
        pr = np.full(4, 0.004); pr[major] = 1 - 0.012
        if rng.random() < 0.15:
            minor = (major + 1 + rng.integers(0, 3)) % 4
            fr = rng.uniform(0.03, 0.5)
            pr[major] -= fr; pr[minor] += fr
        for s in range(S):
            depth = rng.integers(2, 9) if v % 37 == 0 else rng.integers(20, 120)
            data[v, 1 + 4 * s:5 + 4 * s] = rng.multinomial(depth, pr / pr.sum())
    cols = ["Position"] + ["S%d-%s" % (s, b) for s in range(S) for b in "ACGT"]
    return p.DataFrame(data, index=["g%d" % (v // 20) for v in range(V)], columns=cols)


def gen_variant_filter_lrt(out):
    """get_filtered_VariantsLogRatio (Variant_Filter.py:320-390) on a synthetic frame, with and without
    the per-position optimisation, plus the statistical core on the COG0015 base-count sums."""
    import pandas as p
    import desman.Variant_Filter as vf
    frame = lrt_frame()
    rec = dict(frame_values=frame.to_numpy(), frame_index=np.array(frame.index.tolist()),
               frame_columns=np.array(frame.columns.tolist()))
    for tag, opt in (("opt", True), ("noopt", False)):
        f = vf.Variant_Filter(frame, randomState=np.random.RandomState(1), optimise=opt, threshold=3.84,
                              min_coverage=5.0, qvalue_cutoff=1.0e-3)
        f.get_filtered_VariantsLogRatio()
        tm = f.calc_Error_Matrix()
        df = f.selected_variants_todf(frame)
        rec.update({tag + "_ratio": f.ratioNLL, tag + "_pvalue": f.pvalue, tag + "_qvalue": f.qvalue,
                    tag + "_filtered": f.filtered, tag + "_eta": f.eta, tag + "_minV": f.minV, tag + "_tran": tm,
                    tag + "_selected": f.selected, tag + "_snps": np.ascontiguousarray(f.snps_filter),
                    tag + "_selvar_csv": df.to_csv()})
    cog = p.read_csv(os.path.join(REF, "data", "contig_6or16_genesL_scgCOG0015.freq"), header=0, index_col=0)
    f = vf.Variant_Filter(cog, randomState=np.random.RandomState(1), optimise=True, threshold=3.84,
                          min_coverage=5.0, qvalue_cutoff=1.0e-3)
    freq = f.freq.copy()
    f.get_filtered_VariantsLogRatio()
    rec.update(cog_freq=freq, cog_ratio=f.ratioNLL, cog_qvalue=f.qvalue, cog_filtered=f.filtered, cog_eta=f.eta,
               cog_nsel=int(f.NS))
    np.savez_compressed(os.path.join(out, "variant_filter_lrt.npz"), **rec)


def gen_cog(inmft, hsnp, out):
    """Config 1 (COG0015, -g 5 -i 50, default seed): the reference CLI's numeric
    path run through the imported classes (minutes of CPU).  Records fit.txt's
    numbers and the NTF divergence trace.  tau sweep = oracle C restatement."""
    import pandas as p
    import logging
    import desman.Variant_Filter as vf
    logging.basicConfig(level=logging.ERROR)
    variants = p.read_csv(os.path.join(REF, "data", "contig_6or16_genesL_scgCOG0015.freq"),
                          header=0, index_col=0)
    flt = vf.Variant_Filter(variants, randomState=np.random.RandomState(238329), optimise=True,
                            threshold=None, min_coverage=5.0, qvalue_cutoff=1.0e-3)
    seed, G, I = 23724839, 5, 50
    prng = np.random.RandomState(seed)
    cbind.initRNG(); cbind.setRNG(seed)
    nm = inmft.Init_NMFT(flt.snps_filter, G, prng)
    divs = {}
    orig = nm.div_objective
    nm.factorize()
    div_final = orig()
    smp = hsnp.HaploSNP_Sampler(flt.snps_filter, G, prng, max_iter=I)
    smp.tau = np.copy(nm.get_tau(), order='C')
    smp.updateTauIndices()
    smp.gamma = np.copy(nm.get_gamma(), order='C')
    smp.eta = np.copy(flt.eta, order='C')
    tau_init, gamma_init = smp.tau.copy(), smp.gamma.copy()
    smp.update()
    burn_ll = smp.ll_store.copy()
    smp.removeDegenerate()
    smp.update()
    np.savez_compressed(os.path.join(out, "cog0015_g5_i50.npz"), seed=seed, G=G, I=I,
                        V=flt.V, S=flt.S, nmft_div_final=div_final, tau_init=tau_init.astype(np.int8),
                        gamma_init=gamma_init, burn_ll=burn_ll, ll_store=smp.ll_store.copy(),
                        lp_star=smp.lp_star, mean_dev=smp.meanDeviance(), G_final=smp.G,
                        gamma_star=smp.gamma_star.copy(), eta_star=smp.eta_star.copy(),
                        tau_star=smp.tau_star.astype(np.int8), gamma_mean=smp.gammaMean(),
                        note="fit.txt = Fit,%d,%d,%f,%f" % (G, smp.G, smp.lp_star, smp.meanDeviance()))
    print("fit.txt = Fit,%d,%d,%f,%f" % (G, smp.G, smp.lp_star, smp.meanDeviance()))



def gen_gene_assign(out, name="gene_assign", C=7, S=8, G=3, synth_kw=None):
    """desman/GeneAssign.py + desman/Eta_Sampler.py on a synth_genes() data set, twice: through the imported
    classes in main()'s order (intermediate states) and through main() itself (the CSV files it writes).
    The reference needs the pre-1.24 numpy aliases np.int / np.float; they are added to the numpy namespace
    of THIS process only.  tau sweeps = oracle C restatement (GSL-compatible MT19937)."""
    import io
    import logging
    import tempfile
    import pandas as p
    from desman_amd.synth import synth_genes, write_gene_inputs
    np.int, np.float = int, float
    import desman.Eta_Sampler as es
    import desman.GeneAssign as ga

    seed, iters, tau_iter = 4711, 4, 3
    synth_kw = dict(synth_kw or {})
    d = synth_genes(C, S, G, seed=77, **synth_kw)
    rec = dict(C=C, S=S, G=G, seed=seed, iters=iters, tau_iter=tau_iter, synth_seed=77,
               synth_kw=np.array(repr(sorted(synth_kw.items()))))

    class Grab(logging.Handler):
        def __init__(self):
            super().__init__()
            self.lines = []

        def emit(self, r):
            self.lines.append(r.getMessage())

    grab = Grab()
    logging.getLogger().addHandler(grab)
    logging.getLogger().setLevel(logging.INFO)
    with tempfile.TemporaryDirectory() as td:
        paths = write_gene_inputs(d, td)
        scg = p.read_csv(paths[0], header=0, index_col=0)
        gam = p.read_csv(paths[1], header=0, index_col=0)
        cov = p.read_csv(paths[2], header=0, index_col=0)
        eps = p.read_csv(paths[3], header=0, index_col=0).to_numpy()
        var = p.read_csv(paths[4], header=0, index_col=0)
        names = sorted(set(gam.index.values) & set(scg.index.values) & set(cov.columns.values))
        scg, gam = scg.reindex(names), gam.reindex(names)
        gm = gam.to_numpy()
        gm = gm / gm.sum(axis=1)[:, np.newaxis]
        delta = np.multiply(gm, scg['mean'].to_numpy()[:, np.newaxis])
        cov = cov[names]
        prng = np.random.RandomState(seed)
        cbind.initRNG(); cbind.setRNG(seed)
        kl = ga.KLAssign(prng, cov.to_numpy(), delta)
        kl.factorize()
        rec['kl_eta'] = kl.eta.copy()
        tot, acc, full = ga.compGenes(np.rint(kl.eta), d['eta_true'].astype(np.float64))     # pure function
        rec['comp_total'], rec['comp_acc'], rec['comp_matrix'] = tot, acc, full
        rec['kl_div'] = kl.div_objective()
        etaD = np.rint(kl.eta)
        smp = es.Eta_Sampler(prng, var[ga.expand_sample_names(names)], cov, gm, delta, scg['sd'].to_numpy(), eps, etaD,
                             max_iter=iters, max_eta=2, max_var=int(1e10), tau_iter=tau_iter)
        cat = lambda dd: np.concatenate([dd[g] for g in smp.genes if smp.gene_V[g] > 0]).astype(np.int8)
        rec['eta_init'] = smp.eta.copy()
        rec['tau_init'] = cat(smp.gene_tau)
        rec['eta_log_prior'] = smp.eta_log_prior.copy()
        for rnd in (1, 2):
            grab.lines = []
            smp.update()
            rec['eta_store_%d' % rnd] = smp.eta_store.copy()
            rec['eta_star_%d' % rnd] = smp.eta_star.copy()
            rec['gene_ll_%d' % rnd] = smp.gene_ll.copy()
            rec['gene_llstar_%d' % rnd] = smp.gene_llstar.copy()
            rec['ll_%d' % rnd] = smp.ll
            rec['tau_%d' % rnd] = cat(smp.gene_tau)
            rec['ll_log_%d' % rnd] = np.array([float(l.split('nll = ')[1]) for l in grab.lines if l.startswith('Gibbs Iter')])
        smp.restoreFullVariants()
        smp.calcTauStar(smp.eta_star)
        tau_star, tau_mean, pos, contig_index = smp.getTauStar(var)
        rec['tau_star'] = tau_star.astype(np.int8)
        rec['tau_mean_trunc'] = tau_mean.astype(np.int8)          # the reference stores the mean in an int array
        rec['tau_star_ll'] = np.concatenate([smp.gene_ll_tau_star[g] for g in smp.genes])
        rec['tau_store'] = np.concatenate([smp.gene_tau_store[g] for g in smp.genes], axis=1).astype(np.int8)
        rec['pos'] = pos
        rec['contig_index'] = np.array(contig_index)
        # calcTauStar once more, now with a substitute gamma / epsilon (Eta_Sampler.py:397-403: used by the sweeps and their
        # likelihoods, the NMF start keeps the sampler's own gamma): continues both streams from where the first call left them
        g2 = gm * (1.0 + 0.5 * np.cos(np.arange(gm.size).reshape(gm.shape)))
        g2 = g2 / g2.sum(axis=1)[:, np.newaxis]
        e2 = 0.88 * np.eye(4) + 0.03
        smp.calcTauStar(smp.eta_star, gamma=g2, epsilon=e2)
        ts2, tm2, _, _ = smp.getTauStar(var)
        rec['sub_gamma'], rec['sub_epsilon'] = g2, e2
        rec['sub_tau_star'] = ts2.astype(np.int8)
        rec['sub_tau_star_ll'] = np.concatenate([smp.gene_ll_tau_star[g] for g in smp.genes])
        rec['sub_tau_store'] = np.concatenate([smp.gene_tau_store[g] for g in smp.genes], axis=1).astype(np.int8)
        # the same run through main(): output files
        logging.getLogger().removeHandler(grab)
        stub = os.path.join(td, "ga")
        argv = sys.argv
        sys.argv = ["GeneAssign.py", paths[0], paths[1], paths[2], paths[3], "-s", str(seed), "-i", str(iters),
                    "-o", stub, "-v", paths[4], "--assign_tau"]
        try:
            ga.main(sys.argv[1:])
        finally:
            sys.argv = argv
        for suffix in ("etaD_df.csv", "etaS_df.csv", "etaM_df.csv", "eta_df.csv", "_tau_star.csv", "_tau_mean.csv"):
            with open(stub + suffix) as fh:
                rec['file_' + suffix.replace('.csv', '').strip('_')] = np.array(fh.read())
    np.savez_compressed(os.path.join(out, name + ".npz"), **rec)
    print(name, "eta_star", rec['eta_star_2'].tolist(), "ll", rec['ll_2'])



def gen_cog_counts(out):
    """The count tensor of config 1 after the reference's sample filter (Variant_Filter with the CLI's
    defaults): the INPUT of cog0015_g5_i50.npz, so that the recorded reference run can be replayed on the GPU."""
    import pandas as p
    import desman.Variant_Filter as vf
    variants = p.read_csv(os.path.join(REF, "data", "contig_6or16_genesL_scgCOG0015.freq"), header=0, index_col=0)
    flt = vf.Variant_Filter(variants, randomState=np.random.RandomState(238329), optimise=True,
                            threshold=None, min_coverage=5.0, qvalue_cutoff=1.0e-3)
    x = np.ascontiguousarray(flt.snps_filter)
    assert x.max() < 32768 and x.min() >= 0
    np.savez_compressed(os.path.join(out, "cog0015_counts.npz"), counts=x.astype(np.int16), eta0=np.asarray(flt.eta))
    print("cog0015 counts", x.shape, "max", x.max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cog", action="store_true", help="also run config 1 (several minutes)")
    ap.add_argument("--only-cog", action="store_true")
    ap.add_argument("--only-genes", action="store_true")
    ap.add_argument("--only-cog-counts", action="store_true")
    args = ap.parse_args()
    inmft, hsnp, du = import_reference()
    if args.only_cog_counts:
        np.int, np.float = int, float
        gen_cog_counts(HERE)
        return
    if args.only_genes:
        gen_gene_assign(HERE)
        gen_gene_assign(HERE, "gene_assign_lowcov", C=9, S=6, G=4, synth_kw=dict(mean_lo=0.4, mean_hi=2.5, vmax=6))
        return
    if not args.only_cog:
        gen_tau_sweep(hsnp, HERE)
        gen_loglik(hsnp, HERE)
        gen_degenerate(hsnp, HERE)
        gen_gibbs_pieces(hsnp, HERE)
        gen_nmft(inmft, HERE)
        gen_host_formats(HERE)
        gen_resolvenhap(HERE)
        gen_variant_filter_lrt(HERE)
        gen_gene_assign(HERE)
        gen_gene_assign(HERE, "gene_assign_lowcov", C=9, S=6, G=4, synth_kw=dict(mean_lo=0.4, mean_hi=2.5, vmax=6))
    if args.cog or args.only_cog:
        gen_cog(inmft, hsnp, HERE)
        gen_cog_counts(HERE)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
