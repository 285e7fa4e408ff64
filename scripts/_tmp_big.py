import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts, random_state
for V,S,G in [(50000,96,12),(20000,16,5),(20000,40,6)]:
    counts,tt,gg=synth_counts(V,S,G,seed=3)
    tau,gamma,eta=random_state(V,S,G,seed=4)
    ctx=_lib.Context(0); ctx.set_counts(counts)
    ctx.set_state(tau,gamma,eta); ctx.seed(5); ctx.set_timing(True)
    ctx.gibbs_update(5); ctx.sync(); t3=time.time()
    ctx.gibbs_update(20); ctx.sync(); t4=time.time()
    tm=ctx.get_timing()
    print(V,S,G, "ms/iter %.3f"%((t4-t3)/20*1e3), "tau ms %.3f"%(tm['tau'][0]/tm['tau'][1]), ctx.get_trace()["nchange"][-2:])
    ctx.close()
