import numpy as np, sys
r = np.fromfile(sys.argv[1], dtype=np.uint64)
n = int(min(r[6], 65536)); rec = r[8:8+8*n].reshape(n, 8).astype(np.int64)
top = rec[:,1] >> 24; rec[:,1] &= (1 << 24) - 1
xt = rec[:,2] >> 32; rec[:,2] &= (1 << 32) - 1      # part of the K-loop figure spent between the K loop's end and the next tile's first loads (thresholds + wait)
t0 = rec[:,0].min(); start = rec[:,0]-t0; end = rec[:,5]-t0
dur = end-start
print("waves", n, "span", end.max(), "ticks")
print("mean dur", dur.mean(), "prologue", rec[:,1].mean(), "kloop", rec[:,2].mean(), "screen", rec[:,3].mean(), "drain", rec[:,4].mean(), "tile-top", top.mean(), "other", (dur-rec[:,1]-rec[:,2]-rec[:,3]-rec[:,4]-top).mean())
full = (rec[:,7] & 0xffffffff) == 64
em = rec[:,6] & ((1 << 48) - 1); ne = rec[:,6] >> 48
print("eval calls/wave", ne.mean(), "eval distance part", em.mean(), "eval tail", (rec[:,4]-em).mean(), "pairs/wave", (rec[:,7]>>32).mean())
print("full quads: n", full.sum(), "dur", dur[full].mean(), "kloop", rec[full,2].mean(), "evals", (rec[full,7]>>32).mean())
# concurrency over time
ev = np.concatenate([np.stack([start, np.ones(n)],1), np.stack([end, -np.ones(n)],1)])
ev = ev[ev[:,0].argsort()]
lvl = np.cumsum(ev[:,1]); tt = ev[:,0]
tot = end.max(); 
for f in range(10):
    a, b = tot*f/10, tot*(f+1)/10
    m = (tt>=a)&(tt<b)
    print(f"  {f*10:3d}%: avg resident waves {lvl[m].mean() if m.any() else 0:.0f}")
# per quad class (round 3: quads of > 96 queries run in the wide-quad instance)
cntq = rec[:,7] & 0xffffffff
for name, m in (("quads <= 96", cntq <= 96), ("quads > 96 (wide instance)", cntq > 96)):
    if not m.any():
        continue
    d = dur[m]
    print(f"{name}: waves {m.sum()}  first start {start[m].min()}  last end {end[m].max()}  mean dur {d.mean():.0f}  p10/p50/p90 dur "
          f"{np.percentile(d,10):.0f}/{np.percentile(d,50):.0f}/{np.percentile(d,90):.0f}")
    print(f"    prologue {rec[m,1].mean():.0f}  kloop {rec[m,2].mean():.0f} (of which after the loop: {xt[m].mean():.0f})  screen {rec[m,3].mean():.0f}  drain {rec[m,4].mean():.0f}  tile-top {top[m].mean():.0f}"
          f"  other {(d-rec[m,1]-rec[m,2]-rec[m,3]-rec[m,4]-top[m]).mean():.0f}  eval calls/wave {ne[m].mean():.2f}  pairs/wave {(rec[m,7]>>32).mean():.1f}  mean cnt {cntq[m].mean():.0f}")
