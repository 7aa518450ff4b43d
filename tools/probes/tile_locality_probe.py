import sys, numpy as np, scipy.sparse as sp, time
sys.path.insert(0,'/root/repo')
from pose2mesh_release_amd import synth
faces, gL, perm_rev, J = synth.make_graphs('coco')
RMAX, UCAP, ECAP = 32, 120, 896

def merged(L):
    L = sp.csr_matrix(L); L.sort_indices()
    P = (abs(L) + abs(L@L) + sp.identity(L.shape[0])).tocsr()   # pattern of merged row
    P.sort_indices()
    return L, P

def real_rows(L):
    L = L.tocsr(); deg = np.diff(L.indptr)
    iso = (deg == 1) & (L.indices[L.indptr[:-1].clip(max=L.nnz-1)] == np.arange(L.shape[0]))
    return np.where(~iso)[0]

def tiles_of(order, P, shift=0, ucap=UCAP):
    """greedy consecutive tiles in the given row order; returns list of (rows, union size, entries)"""
    out=[]; i=0; n=len(order)
    while i<n:
        uni=set(); rows=0; ent=0
        while i+rows<n and rows<RMAX:
            r=order[i+rows]; cols=P.indices[P.indptr[r]:P.indptr[r+1]]>>shift
            nu=uni|set(cols.tolist())
            if rows>0 and (len(nu)>ucap or ent+len(cols)>ECAP): break
            uni=nu; ent+=len(cols); rows+=1
        out.append((rows,len(uni),ent)); i+=rows
    return out

def stats(name, tl):
    r=np.array([t[0] for t in tl]); u=np.array([t[1] for t in tl])
    print(f"{name:28s} tiles {len(tl):5d} rows/tile {r.mean():5.1f} union/row {u.sum()/r.sum():5.2f} union mean {u.mean():6.1f} max {u.max()} <=96: {(u<=96).mean():.2f} <=112: {(u<=112).mean():.2f}")

def grow_order(P1, real, seed_rule="adjacent"):
    """greedy patch growing over the 1-ring graph restricted to real vertices: patches of 32, each grown by taking the
    frontier vertex with most neighbours already in the patch; next seed = unassigned vertex adjacent to assigned region with
    max assigned neighbours (keeps patches packed)."""
    import heapq
    n=P1.shape[0]; isreal=np.zeros(n,bool); isreal[real]=True
    assigned=np.zeros(n,bool); order=[]
    indptr,indices=P1.indptr,P1.indices
    nb=lambda v: indices[indptr[v]:indptr[v+1]]
    cnt_assigned=np.zeros(n,int)   # number of assigned neighbours
    remaining=set(real.tolist())
    seed=real[0]
    boundary=set()
    while remaining:
        if seed is None:
            # pick unassigned real vertex with most assigned neighbours (from boundary), else any
            cand=[v for v in boundary if not assigned[v]]
            if cand:
                seed=max(cand,key=lambda v:(cnt_assigned[v],-v))
            else:
                seed=min(remaining)
        patch=[]; inpatch={}
        heap=[(-0,seed)]; score={seed:0}
        while heap and len(patch)<RMAX:
            s,v=heapq.heappop(heap)
            if assigned[v] or v in inpatch or -s!=score.get(v,0): 
                if assigned[v] or v in inpatch: continue
            inpatch[v]=1; patch.append(v)
            for w in nb(v):
                if isreal[w] and not assigned[w] and w not in inpatch:
                    score[w]=score.get(w,0)+1
                    heapq.heappush(heap,(-score[w],w))
        for v in patch:
            assigned[v]=True; remaining.discard(v); boundary.discard(v)
            for w in nb(v):
                cnt_assigned[w]+=1
                if isreal[w] and not assigned[w]: boundary.add(w)
        order.extend(patch); seed=None
    return np.array(order)

for lvl in range(0,5):
    L,P=merged(gL[lvl]); real=real_rows(L)
    V=L.shape[0]
    print(f"--- level {lvl} V={V} real={len(real)} nnz merged/row {P[real].nnz/len(real):.1f}")
    stats("tree order (current)", tiles_of(real,P))
    P1=(abs(L)+sp.identity(V)).tocsr(); P1.sort_indices()
    t=time.time(); o=grow_order(P1,real); 
    assert sorted(o.tolist())==sorted(real.tolist())
    stats("greedy grow (1-ring)", tiles_of(o,P)); 
    stats("greedy grow, shift1 union", tiles_of(o,P,1))
    stats("tree order, shift1 union", tiles_of(real,P,1))
    print("   grow time %.2fs"%(time.time()-t))

def grow_min_union(P, P1, real):
    """patch growing that adds the frontier vertex whose merged row adds the fewest NEW union columns"""
    n=P.shape[0]; isreal=np.zeros(n,bool); isreal[real]=True
    assigned=np.zeros(n,bool); order=[]
    rows=[set(P.indices[P.indptr[v]:P.indptr[v+1]].tolist()) for v in range(n)]
    nb=lambda v: P1.indices[P1.indptr[v]:P1.indptr[v+1]]
    remaining=set(real.tolist()); boundary=set(); seed=real[0]
    cnt_assigned=np.zeros(n,int)
    while remaining:
        if seed is None:
            cand=[v for v in boundary if not assigned[v]]
            seed=max(cand,key=lambda v:(cnt_assigned[v],-v)) if cand else min(remaining)
        patch=[seed]; uni=set(rows[seed]); front=set(w for w in nb(seed) if isreal[w] and not assigned[w] and w!=seed)
        inp={seed}
        while len(patch)<RMAX and front:
            best=min(front,key=lambda w:(len(rows[w]-uni),w))
            if len(uni|rows[best])>UCAP: break
            front.discard(best); inp.add(best); patch.append(best); uni|=rows[best]
            for w in nb(best):
                if isreal[w] and not assigned[w] and w not in inp: front.add(w)
        for v in patch:
            assigned[v]=True; remaining.discard(v); boundary.discard(v)
            for w in nb(v):
                cnt_assigned[w]+=1
                if isreal[w] and not assigned[w]: boundary.add(w)
        order.extend(patch); seed=None
    return np.array(order)

print("=== min-union growth")
for lvl in (0,2):
    L,P=merged(gL[lvl]); real=real_rows(L); V=L.shape[0]
    P1=(abs(L)+sp.identity(V)).tocsr(); P1.sort_indices()
    o=grow_min_union(P,P1,real)
    assert sorted(o.tolist())==sorted(real.tolist())
    stats(f"lvl{lvl} min-union growth", tiles_of(o,P))
