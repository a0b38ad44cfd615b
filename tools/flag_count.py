import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from densematcher_amd import synth
from densematcher_amd.engine import MatchEngine
eng=MatchEngine(0)
n,D=2048,768
for sigma in (0.1,1.0):
    F1,F2,_=synth.feature_pair(n,n,D,1000,2000,sigma=sigma)
    nn,best,margin=eng.simnn(F2[None],F1[None],return_scores=True)
    m=margin[0].cpu().numpy(); b=best[0].cpu().numpy()
    tau=2*1.01*D*(1+1/16)*1.1920929e-7
    print('sigma',sigma,'best mean',b.mean(),'margin median',np.median(m),'min',m.min(),'flagged frac',(m<=tau).mean(),'tau',tau)
