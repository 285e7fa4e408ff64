import sys; sys.path.insert(0,'.')
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts, random_state
V,S,G=10000,64,8
counts,tt,gg=synth_counts(V,S,G,1234)
tau,gamma,eta=random_state(V,S,G,seed=1)
ctx=_lib.Context(0); ctx.set_counts(counts); ctx.set_state(tau,gamma,eta); ctx.seed(1)
n=int(sys.argv[1]) if len(sys.argv)>1 else 10
ctx.gibbs_update(n)
print(ctx.get_trace()["ll"][-1])
