"""Host loader with the constructor semantics of desman/Variant_Filter.py.

Only what the Gibbs path needs: reshape the CSV frame into
snps[V,S,4], drop low-coverage samples, default eta, selection bookkeeping and
`select_Random`.  The likelihood-ratio variant filter (`-f`,
Variant_Filter.py:320-390) is upstream of the hot path and NOT implemented
(SURVEY sec. 8 row f3): asking for it raises.
"""
import numpy as np


class Variant_Filter:

    def __init__(self, variants, randomState, optimise=True, threshold=3.84, min_coverage=5.0,
                 qvalue_cutoff=0.1, max_iter=100, min_p=0.01, mCogFilter=2.0, cogSampleFrac=0.95,
                 Nthreshold=10):
        m = variants.to_numpy()
        self.genes = list(variants.index)
        self.position = m[:, 0]                                    # first data column = Position (:75-77)
        m = np.delete(m, 0, 1)
        snps = np.reshape(m, (m.shape[0], m.shape[1] // 4, 4))
        self.randomState = randomState
        depth = snps.sum(axis=2)
        self.sample_filter = np.mean(depth, axis=0) > min_coverage   # (:87)
        self.sample_indices = np.where(self.sample_filter)[0].tolist()
        self.snps_filter = snps[:, self.sample_filter, :]
        self.V, self.S = self.snps_filter.shape[0], self.snps_filter.shape[1]
        self.freq = self.snps_filter.sum(axis=1)
        self.ffreq = self.freq.astype(np.float64)
        self.threshold = threshold
        self.qvalue_cutoff = qvalue_cutoff
        self.optimise = optimise
        self.filtered = np.zeros(self.V, dtype=bool)
        self.max_iter = max_iter
        self.eta = 0.96 * np.identity(4) + 0.01 * np.ones((4, 4))    # (:108)
        self.upperP = 1.0 - min_p
        self.NS = self.V
        self.selected = np.ones(self.V, dtype=bool)
        self.selected_indices = np.where(self.selected)[0].tolist()
        self.randomSelect = False

    def get_filtered_VariantsLogRatio(self):
        raise NotImplementedError(
            "the likelihood-ratio variant filter (-f) is outside the accelerated hot path (SURVEY sec. 8, row f3); "
            "run the reference's Variant_Filter.py as a pre-pass and feed its sel_var.csv")

    def select_Random(self, random_select):
        """sorted choice without replacement from the filter's RandomState (:392-410)."""
        if random_select < self.NS:
            self.randomSelect = True
            select = np.sort(self.randomState.choice(self.NS, random_select, replace=False))
            self.snps_filter_original = np.copy(self.snps_filter)
            self.snps_filter = self.snps_filter[select, :, :]
            self.NS = random_select
            self.selected_indices_original = np.copy(self.selected_indices)
            self.selected_indices = [self.selected_indices[i] for i in select]
            self.selected_original = np.copy(self.selected)
            self.selected = np.zeros(self.V, dtype=bool)
            self.selected[self.selected_indices] = True
        return self.snps_filter
