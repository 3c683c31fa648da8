import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, pq_vector_amd as pqv
from oracle_binding import Oracle
o=Oracle()
n,dim,kc,k,nprobe=5000,64,20,100,6
rng=np.random.default_rng(n*31+dim)
data=rng.random((n,dim),dtype=np.float32)
oidx=o.build_index(data,n_clusters=kc,workers=1,max_iters=5)
queries=rng.random((9,dim),dtype=np.float32)
c=pqv.Corpus.upload(data); s=pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()),c)
for kk in (64,65,100,128,129,256,300):
    rows,dist,nf,nc=s.topk(queries,kk,nprobe,sqrt_out=False)
    # brute: oracle candidates + exact d2
    bad=0
    for q in range(9):
        cand=oidx.candidate_rows(queries[q],nprobe)
        d2=np.array([o.l2_ref4(queries[q],data[r]) for r in cand],np.float32)
        order=np.lexsort((np.arange(len(cand)),d2.view(np.uint32)))[:kk]
        exp_rows=cand[order]; exp_d=d2[order]
        m=int(nf[q])
        if m!=len(order) or not (rows[q,:m]==exp_rows).all() or not (dist[q,:m].view(np.uint32)==exp_d.view(np.uint32)).all():
            bad+=1
            if bad==1:
                neq=np.nonzero(rows[q,:m]!=exp_rows[:m])[0]
                print("k",kk,"q",q,"m",m,"first mismatches",neq[:10], rows[q,neq[:5]], exp_rows[neq[:5]], dist[q,neq[:5]], exp_d[neq[:5]])
                print(" missing", set(exp_rows.tolist())-set(rows[q,:m].tolist()), "extra", set(rows[q,:m].tolist())-set(exp_rows.tolist()))
    print("k",kk,"bad queries",bad)
