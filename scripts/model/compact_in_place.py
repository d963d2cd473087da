"""Model of the in-place compaction (limitador_amd/csrc/rl_kernels.hpp: k_compact_bounds + k_compact_seg) — the same walk, slot
for slot: segment bounds = the first EMPTY slot at or after every SEG-th slot of the UNMODIFIED table; a wave owns the clusters
that start inside its segment and walks them in 64-slot steps through a window (64 + EXTRA slots read ahead); a cluster without
room in its first 64 slots for what lies behind them takes the slow path (the table itself as state).  Checked: no tombstone is
left, exactly the live keys remain, every one is reachable from its home slot by linear probing.  tests/test_compact_algorithm_cpu.py
runs it; `python scripts/model/compact_in_place.py` runs 400 cases."""
import random
EMPTY, TOMB = -1, -2
SEG=256; EXTRA=16
def build(cap, nkeys, ndel, seed):
    rnd = random.Random(seed); home={}; t=[EMPTY]*cap
    keys=list(range(nkeys))
    for k in keys:
        h = rnd.randrange(cap) if rnd.random()<0.7 else rnd.randrange(max(1,cap//8))
        home[k]=h; s=h
        while t[s]!=EMPTY: s=(s+1)%cap
        t[s]=k
    dels=rnd.sample(keys, ndel)
    for k in dels: t[t.index(k)]=TOMB
    live=set(keys)-set(dels)
    for k in range(nkeys, nkeys+nkeys//10):
        h=rnd.randrange(cap); home[k]=h; s=h
        while t[s]!=EMPTY: s=(s+1)%cap
        t[s]=k; live.add(k)
    return t,home,live
def walk(t,cap,home,a,base,win):
    def rd(sl):
        r=sl-base
        return win[r] if r<64+EXTRA else t[sl%cap]
    freem=0; d=0; ended=False
    while d<64:
        tag=rd(a+d)
        if tag==EMPTY: ended=True; break
        if tag==TOMB: freem|=1<<d; d+=1; continue
        hd=(home[tag]-a)%cap; assert hd<=d
        cand=(freem>>hd)<<hd
        if cand:
            q=(cand&-cand).bit_length()-1; t[(a+q)%cap]=tag; freem=(freem&~(1<<q))|(1<<d)
        d+=1
    f=freem
    while f:
        q=(f&-f).bit_length()-1; t[(a+q)%cap]=EMPTY; f&=f-1
    if not ended:
        while d<cap:
            tag=rd(a+d)
            if tag==EMPTY: break
            if tag==TOMB: t[(a+d)%cap]=EMPTY; d+=1; continue
            q=(home[tag]-a)%cap
            while q<d and t[(a+q)%cap]!=EMPTY: q+=1
            if q<d: t[(a+q)%cap]=tag; t[(a+d)%cap]=EMPTY
            d+=1
    return a+d
def run(cap,nkeys,ndel,seed):
    t,home,live=build(cap,nkeys,ndel,seed)
    nseg=max(cap//SEG,1)
    bounds=[]
    for w in range(nseg):
        s=w*SEG
        while t[s%cap]!=EMPTY: s+=1
        bounds.append(s)
    order=list(range(nseg)); random.Random(seed).shuffle(order)   # waves in any order
    for gw in order:
        lo=bounds[gw]; hi=bounds[gw+1] if gw+1<nseg else bounds[0]+cap
        carry=lo; prev_ne=False
        nxt=[t[(lo+1+i)%cap] for i in range(64+EXTRA)] if lo+1<hi else None
        base=lo+1
        while base<hi:
            win=nxt
            if base+64<hi: nxt=[t[(base+64+i)%cap] for i in range(64+EXTRA)]
            ne=[win[l]!=EMPTY for l in range(64)]
            ends=[]
            for l in range(64):
                slot=base+l
                b=ne[l-1] if l else prev_ne
                if slot-1==carry: b=False
                if ne[l] and not b and slot>carry and slot<hi:
                    ends.append(walk(t,cap,home,slot,base,win))
            if ends: carry=ends[-1]
            prev_ne=ne[63]
            base+=64
    assert TOMB not in t
    present=[x for x in t if x!=EMPTY]
    assert sorted(present)==sorted(live),(len(present),len(live))
    for k in live:
        s=home[k]
        while t[s]!=k:
            assert t[s]!=EMPTY,("unreachable",k); s=(s+1)%cap
def case(seed):
    cap=random.Random(seed).choice([64,128,256,1024,2048])
    load=random.Random(seed+1).uniform(0.2,0.85)
    n=int(cap*load); run(cap,n,random.Random(seed+2).randrange(0,n+1),seed)
if __name__ == "__main__":
    for seed in range(400): case(seed)
    print("ok")
