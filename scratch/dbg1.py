import sys; sys.path.insert(0,'.')
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts, random_state
from oracle import cbind
V,S,G=1000,16,5
counts,_,_=synth_counts(V,S,max(G,2),seed=V+S)
tau,gamma,eta=random_state(V,S,G,seed=G)
ctx=_lib.Context(0); ctx.set_counts(counts); ctx.set_state(tau,gamma,eta); ctx.seed(4242)
mt=cbind.MT19937(4242); ref=tau.copy()
for sw in range(3):
    u=mt.uniform(V*G)
    n_ref,lr=cbind.sample_tau_u(ref,gamma,eta,counts,u,want_logp=True)
    n,l=ctx.sample_tau(want_logp=True)
    got,_,_=ctx.get_state()
    d=np.argwhere(np.argmax(got,2)!=np.argmax(ref,2))
    print(sw,n,n_ref,len(d), d[:5].tolist(), np.abs(l-lr).max())
    if len(d):
        v,g=d[0]
        print(l[v], lr[v], u[v*G:(v+1)*G])
    ref=got.copy()
