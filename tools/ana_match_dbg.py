#!/usr/bin/env python3
"""Per-wavefront timeline of hamming_mfma_k from the stamps the library writes with MSLAM_MATCH_DEBUG=<file> (eighth call of the
process): wavefront lifetimes, start times, concurrency over the kernel's span, wavefronts per CU.

usage: MSLAM_MATCH_DEBUG=gpurun_out/match_dbg.txt python tools/match_probe.py && python tools/ana_match_dbg.py gpurun_out/match_dbg.txt"""
import sys, numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.uint64)
t0, t1, flag, hw = d[:,1].astype(np.int64), d[:,2].astype(np.int64), d[:,3], d[:,4]
act = (flag & 1) == 1
xcc = (flag >> 8) & 0xF
base = t0[act].min()
print("waves", len(d), "active", act.sum(), "span (100 MHz ticks)", (t1[act].max()-base), "=", (t1[act].max()-base)/100.0, "us")
life = (t1-t0)[act]/100.0
print("active wave life us: mean %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f" % (life.mean(), *np.percentile(life,[10,50,90]), life.max()))
st = (t0[act]-base)/100.0
print("start times us: p10 %.2f p50 %.2f p90 %.2f max %.2f" % (*np.percentile(st,[10,50,90]), st.max()))
# concurrency over time
ev = np.concatenate([np.stack([t0[act]-base, np.ones(act.sum())],1), np.stack([t1[act]-base, -np.ones(act.sum())],1)])
ev = ev[np.argsort(ev[:,0], kind='stable')]
cur=0; last=0; acc={}
T = ev[-1,0]
bins = np.zeros(20); 
for t,dv in ev:
    b0=int(last/T*20); b1=int(t/T*20)
    # accumulate cur*(t-last) into bins (approx: all to bin of last)
    bins[min(b0,19)] += cur*(t-last)
    cur+=dv; last=t
print("avg concurrent active waves per 5% of the span:", np.round(bins/(T/20)).astype(int))
se = (hw >> 13) & 7; cu = (hw >> 8) & 15; simd = (hw >> 4) & 3
key = xcc[act]*10000 + se[act]*100 + cu[act]
u, c = np.unique(key, return_counts=True)
print("distinct (xcc,se,cu):", len(u), "waves per CU min/mean/max", c.min(), c.mean(), c.max())
ux, cx = np.unique(xcc[act], return_counts=True); print("active waves per XCC", dict(zip(ux.tolist(), cx.tolist())))
