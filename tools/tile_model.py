#!/usr/bin/env python3
"""Offline cost model of the flow search's query tiling: for a
KITTI-sized frame pair, evaluated lane-candidate pairs, in-window pairs and VALU
instructions per tiling scheme, with the per-candidate instruction counts of the
shipped kernel (9.25 without accept test, 12.25 with the v-only test, 13.25 with the
full test).  CPU only (uses the oracle for the features)."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg=g.load_package(); ob=g.load_oracle(); o=ob.Oracle(); p=ob.Params.default()
W,H=1241,376; dims=[W,H,1248]; bs=50; r=200
ubn=-(-W//bs); vbn=-(-H//bs)
fq=o.compute_features(p,pkg.synth.frame(W,H,5,1),dims)[1]
fc=o.compute_features(p,pkg.synth.frame(W,H,0,0),dims)[1]
def binned(f):
    c=f[:,3]; ub=np.minimum(f[:,0]//bs,ubn-1); vb=np.minimum(f[:,1]//bs,vbn-1)
    return c,ub,vb
cq,uq,vq=binned(fq); cc,uc,vc=binned(fc)
# candidate counts per (class, ub, vb)
cnt=np.zeros((4,ubn,vbn),np.int64)
np.add.at(cnt,(cc,uc,vc),1)
COST={0:9.25,1:12.25,2:13.25}
def tile_cost(idx):
    """idx: indices of queries (same class) in one tile (<=64) -> (wave_ops, evaluated_cands, inwindow_pairs)"""
    c=cq[idx[0]]; u=fq[idx,0]; v=fq[idx,1]
    ulo,uhi,vlo,vhi=u-r,u+r,v-r,v+r
    UB0=min(np.minimum(np.maximum(ulo,0)//bs,ubn-1)); UB1=max(np.minimum(np.maximum(uhi,0)//bs,ubn-1))
    VB0=min(np.minimum(np.maximum(vlo,0)//bs,vbn-1)); VB1=max(np.minimum(np.maximum(vhi,0)//bs,vbn-1))
    ULO_MAX=ulo.max(); UHI_MIN=uhi.min(); VLO_MAX=vlo.max(); VHI_MIN=vhi.min()
    VA0=max(VB0,(max(VLO_MAX,0)+bs-1)//bs); VA1=min(VB1,(VHI_MIN+1)//bs-1)
    ops=0; ev=0
    for ub in range(UB0,UB1+1):
        interior = ub*bs>=ULO_MAX and ub*bs+bs-1<=UHI_MIN
        for vb in range(VB0,VB1+1):
            n=cnt[c,ub,vb]
            t = 2 if not interior else (0 if VA0<=vb<=VA1 else 1)
            ops+=n*COST[t]; ev+=n
    # in-window pairs
    m=fc[cc==c]
    inw=((np.abs(m[None,:,0]-u[:,None])<=r)&(np.abs(m[None,:,1]-v[:,None])<=r)).sum()
    return ops,ev,inw
def evaluate(tiles,name):
    ops=ev=inw=0; lanes=0
    for t in tiles:
        a,b,c_=tile_cost(t); ops+=a; ev+=b; inw+=c_; lanes+=len(t)
    nt=len(tiles)
    print(f"{name:28s} tiles {nt:4d} avg fill {lanes/nt:5.1f}  wave-ops {ops:10.0f}  evaluated lane-pairs {ev*64:.3e}  in-window {inw:.3e}  eff {inw/(ev*64):.3f}  ops per in-window pair {ops*64/inw:.2f}")
# current tiling: bin order (c, ub, vb), 64 consecutive per class
order=np.lexsort((np.arange(len(fq)),vq,uq,cq))
tiles=[]
for c in range(4):
    idx=order[cq[order]==c]
    tiles+= [idx[i:i+64] for i in range(0,len(idx),64)]
evaluate(tiles,"current (64 in bin order)")
# 2-D blocks: ku columns x kv bins
for ku,kv in ((1,4),(2,2),(2,3),(3,2),(1,8),(2,4),(4,1),(3,3)):
    tiles=[]
    for c in range(4):
        for U in range(0,ubn,ku):
            for V in range(0,vbn,kv):
                idx=np.flatnonzero((cq==c)&(uq>=U)&(uq<U+ku)&(vq>=V)&(vq<V+kv))
                # sort inside the block by (vb, ub) to keep sub-tiles compact
                tiles+=[idx[i:i+64] for i in range(0,len(idx),64)]
    evaluate([t for t in tiles if len(t)],f"blocks {ku}x{kv} bins")
print("---- linear orders, full tiles of 64")
for k in (1,2,3,4,5,6):
    order=np.lexsort((np.arange(len(fq)),uq%k,vq,uq//k,cq))
    tiles=[]
    for c in range(4):
        idx=order[cq[order]==c]
        tiles+=[idx[i:i+64] for i in range(0,len(idx),64)]
    evaluate(tiles,f"column groups of {k}, v-major")
# Hilbert-ish: sort by vb//2 bands? order (c, vb//kv, ub, vb%kv)
for kv in (2,3,4):
    order=np.lexsort((np.arange(len(fq)),vq%kv,uq,vq//kv,cq))
    tiles=[]
    for c in range(4):
        idx=order[cq[order]==c]
        tiles+=[idx[i:i+64] for i in range(0,len(idx),64)]
    evaluate(tiles,f"v bands of {kv} bins, u-major")
print("---- tile size")
order=np.lexsort((np.arange(len(fq)),vq,uq,cq))
for T in (16,32,64,128):
    tiles=[]
    for c in range(4):
        idx=order[cq[order]==c]
        tiles+=[idx[i:i+T] for i in range(0,len(idx),T)]
    ops=ev=inw=0
    for t in tiles:
        a,b,c_=tile_cost(t); ops+=a*T/64; ev+=b*T; inw+=c_
    print(f"tile {T:3d}: eff {inw/ev:.3f}  lane-ops per in-window pair {ops*64/inw:.2f}")
