import sys, os, time
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'l-giremi_amd'))
import numpy as np, lgmi
from lgmi.synth import footprint_blocks
hb = footprint_blocks(20000, seed=20250810, cache_dir='/tmp')
eng = lgmi.Engine(0)
res = eng.run(hb, min_common=6, het_only=True, n_shuffles=1000, seed=1, emit_counts=True)
c = res.row_counts.reshape(-1,3,3).astype(np.int64)
R = c.sum(axis=2); C = c.sum(axis=1)
gen = ((R>0).sum(axis=1)>2) | ((C>0).sum(axis=1)>2)
print('rows', len(c), 'general', int(gen.sum()))
Rg, Cg = np.sort(R[gen],axis=1), np.sort(C[gen],axis=1)
nt = np.ones(gen.sum())
for a in (0,1):
    for b in (0,1):
        nt *= np.minimum(Rg[:,a], Cg[:,b]) + 1
en = nt <= 4000
print('enumerable', int(en.sum()), 'mean tables', nt[en].mean(), 'median', np.median(nt[en]), 'sum tables', nt[en].sum())
N = R[gen].sum(axis=1)
print('non-enumerable', int((~en).sum()), 'N quantiles', np.quantile(N[~en],[0.1,0.5,0.9,0.99]))
print('nt quantiles of non-enumerable', np.quantile(nt[~en],[0.1,0.5,0.9]))
for cap in (8192, 16384, 65536, 262144, 1<<20):
    print('cap', cap, 'enumerable', int((nt<=cap).sum()), 'sum tables %.3g' % nt[nt<=cap].sum())
eng.close()
