"""the whole Gibbs iteration on the device against the oracle composition (tests/test_gpu_parity.py:
test_gibbs_update_is_self_consistent_with_oracle) over random shapes and the boundaries of the kernel selection (S = 16/17, 32/33, 48,
64/65, 96/97, 128/129, 200; G = 1, 2, 9/10, 12, 16/17), all three mu/E specifications.  usage: fuzz_gibbs.py [n_random] [seed]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from desman_amd import _lib
import test_gpu_parity as tp
n_rand = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
shapes = [(150, S, 4) for S in (1, 2, 16, 17, 32, 33, 48, 64, 65, 96, 97, 128, 129, 192, 193, 200, 256, 257, 300, 384, 385, 512)] + [(260, 24, G) for G in (1, 2, 9, 10, 12, 16, 17)] + [(V, 64, 8) for V in (1, 3, 64, 257)]
shapes += [(int(rs.randint(1, 700)), int(rs.randint(1, 140)), int(rs.randint(1, 13))) for _ in range(n_rand)]
bad = 0
for V, S, G in shapes:
    for spec in (2, 3, 1):
        if G > 16 and spec >= 2: continue              # the aggregated specifications stop at 16 haplotypes
        ctx = _lib.Context(0)
        try:
            tp.test_gibbs_update_is_self_consistent_with_oracle.__wrapped__(ctx, V, S, G, 3, spec) if hasattr(tp.test_gibbs_update_is_self_consistent_with_oracle, "__wrapped__") else tp.test_gibbs_update_is_self_consistent_with_oracle(ctx, V, S, G, 3, spec)
        except Exception as e:
            bad += 1
            print("FAIL V=%d S=%d G=%d spec=%d: %s" % (V, S, G, spec, str(e).splitlines()[0][:200] if str(e) else type(e).__name__), flush=True)
        finally:
            ctx.close()
print("shapes %d x 3 specs, failures %d" % (len(shapes), bad))
