#!/usr/bin/env python3
"""How good is a tree built on the device?  A host model (no GPU) behind k_lbvh.hip's design (DESIGN.md section 4 "Host side", round 5).

Over the dungeon's triangles (the engine's own device stream) four 4-wide trees are built and walked by the same rays (primary rays of a
48 x 32 camera, one uniform-hemisphere bounce per hit pixel), with the product's rules (nearest child first, the rest pushed):
  sah     the host's binned-SAH tree collapsed top-down by surface area (st_bvh_refresh.cpp build_wide_topology: what ships)
  karras  the binary radix tree of Karras 2012 over 30-bit Morton codes (+ index tie-break), single-triangle leaves, collapsed the same way
          (what k_lbvh.hip builds)
  ploc8   parallel locally-ordered clustering, radius 8, collapsed the same way
  implicit an equal-count 4-ary tree over the Morton order (runs of R triangles): the obvious implicit tree — 2.5x the steps, dropped
Prints node and triangle steps per ray and the deepest stack of the bounce rays.

    python tools/lbvh_sim.py [subdivide] [R]
"""
import sys, math, os
sys.setrecursionlimit(100000)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strolle_amd import Engine, scenes
sub = int(sys.argv[1]) if len(sys.argv) > 1 else 0
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2
e = Engine(device=-1); scenes.build_dungeon(e, subdivide=sub); e.tick()
S = e.read_scene(4).reshape(-1, 4, 4).astype(np.float32); SU = S.view(np.uint32)
is_int = SU[:, 0, 3] == 0
F = float(np.float32(3.4e38))
# ---- SAH wide (as tools/bvh4_sim.py)
def area(lo, hi):
    d = np.maximum(hi - lo, 0); return d[0]*d[1]+d[1]*d[2]+d[2]*d[0]
def ch2(k):
    far = int(SU[k,1,3]//64); return [(S[k,0,:3],S[k,1,:3],('n',k+1) if is_int[k+1] else ('l',k+1)), (S[k,2,:3],S[k,3,:3],('n',far) if is_int[far] else ('l',far))]
sah = {}
todo=[0]
while todo:
    k=todo.pop(); ch=ch2(k)
    while len(ch)<4:
        c=[(area(x[0],x[1]),i) for i,x in enumerate(ch) if x[2][0]=='n']
        if not c: break
        _,i=max(c); x=ch.pop(i); ch[i:i]=ch2(x[2][1])
    sah[k]=ch; todo+=[x[2][1] for x in ch if x[2][0]=='n']
def sah_leaf(k):
    out=[k]
    while SU[out[-1],0,0]&1: out.append(out[-1]+1)
    return out
# ---- triangles
leaves = np.flatnonzero(~is_int)
P0,E1,E2 = S[leaves,1,:3],S[leaves,2,:3],S[leaves,3,:3]
V = np.stack([P0,P0+E1,P0+E2],1); lo=V.min(1); hi=V.max(1); cen=(lo+hi)*0.5
smin,smax=cen.min(0),cen.max(0)
q=np.clip(((cen-smin)/np.maximum(smax-smin,1e-20)*1023).astype(np.int64),0,1023)
def part(x):
    x=x&0x3ff; x=(x|(x<<16))&0x30000ff; x=(x|(x<<8))&0x300f00f; x=(x|(x<<4))&0x30c30c3; x=(x|(x<<2))&0x9249249; return x
m=part(q[:,0])|(part(q[:,1])<<1)|(part(q[:,2])<<2)
order=np.argsort(m,kind='stable'); N=len(order)
lo_s,hi_s=lo[order],hi[order]
M=(N+R-1)//R
run_lo=np.array([lo_s[i*R:(i+1)*R].min(0) for i in range(M)]); run_hi=np.array([hi_s[i*R:(i+1)*R].max(0) for i in range(M)])
levels=[]  # level 0: parents of runs
clo,chi,n=run_lo,run_hi,M
kind='l'
tree={}  # (h,j) -> children
h=0
while True:
    cnt=(n+3)//4
    nlo=np.array([clo[4*j:4*j+4].min(0) for j in range(cnt)]); nhi=np.array([chi[4*j:4*j+4].max(0) for j in range(cnt)])
    for j in range(cnt):
        tree[(h,j)]=[(clo[c],chi[c],(kind,(h-1,c) if kind=='n' else c)) for c in range(4*j,min(4*j+4,n))]
    clo,chi,n,kind=nlo,nhi,cnt,'n'; h+=1
    if cnt==1: break
root=(h-1,0)
implicit_tree, implicit_root = tree, root
def sa(lo_,hi_):
    d=np.maximum(hi_-lo_,0); return d[...,0]*d[...,1]+d[...,1]*d[...,2]+d[...,2]*d[...,0]
# binary tree as arrays: nodes: (lo,hi,left,right) ; leaves negative ids -> triangle index (sorted position)
class BT:
    def __init__(s): s.lo=[];s.hi=[];s.l=[];s.r=[]
    def add(s,lo_,hi_,l,r): s.lo.append(lo_);s.hi.append(hi_);s.l.append(l);s.r.append(r); return len(s.lo)-1
def karras(codes):
    bt=BT()
    def build(a,b):  # [a,b) sorted positions
        if b-a==1: return -(a+1)
        c0,c1=int(codes[a]),int(codes[b-1])
        if c0==c1: mid=(a+b)//2
        else:
            bit=(c0^c1).bit_length()-1
            # first position whose code has that bit set
            lo_,hi_=a,b-1
            mask=~((1<<bit)-1)
            pref=c1&mask
            while lo_<hi_:
                m_=(lo_+hi_)//2
                if (int(codes[m_])&mask)>=pref: hi_=m_
                else: lo_=m_+1
            mid=lo_
        l=build(a,mid); r=build(mid,b)
        def box(x):
            if x<0: i=-x-1; return lo_s[i],hi_s[i]
            return bt.lo[x],bt.hi[x]
        (llo,lhi),(rlo,rhi)=box(l),box(r)
        return bt.add(np.minimum(llo,rlo),np.maximum(lhi,rhi),l,r)
    root=build(0,len(codes)); return bt,root
def ploc(radius):
    bt=BT()
    ids=[-(i+1) for i in range(N)]; clo=lo_s.copy(); chi=hi_s.copy()
    while len(ids)>1:
        n=len(ids); best=np.full(n,np.inf); nb=np.full(n,-1)
        for off in range(1,radius+1):
            if off>=n: break
            a_lo=np.minimum(clo[:-off],clo[off:]); a_hi=np.maximum(chi[:-off],chi[off:]); ar=sa(a_lo,a_hi)
            idx=np.arange(n-off)
            upd=ar<best[idx]; best[idx[upd]]=ar[upd]; nb[idx[upd]]=idx[upd]+off
            idx2=idx+off
            upd=ar<best[idx2]; best[idx2[upd]]=ar[upd]; nb[idx2[upd]]=idx[upd]
        merged=np.zeros(n,bool); nids=[];nlo=[];nhi=[]
        for i in range(n):
            j=nb[i]
            if merged[i]: continue
            if j>=0 and nb[j]==i and not merged[j] and i<j:
                k=bt.add(np.minimum(clo[i],clo[j]),np.maximum(chi[i],chi[j]),ids[i],ids[j]); merged[j]=True
                nids.append(k);nlo.append(bt.lo[k]);nhi.append(bt.hi[k])
            elif not (j>=0 and nb[j]==i and i>j):
                nids.append(ids[i]);nlo.append(clo[i]);nhi.append(chi[i])
        ids=nids;clo=np.array(nlo);chi=np.array(nhi)
    return bt,ids[0]
def collapse(bt,root):
    tr={}
    def kids(x):
        out=[]
        for c in (bt.l[x],bt.r[x]):
            if c<0: i=-c-1; out.append((lo_s[i],hi_s[i],('l',i)))
            else: out.append((bt.lo[c],bt.hi[c],('n',c)))
        return out
    todo=[root]
    while todo:
        k=todo.pop(); ch=kids(k)
        while len(ch)<4:
            c=[(float(sa(x[0],x[1])),i) for i,x in enumerate(ch) if x[2][0]=='n']
            if not c: break
            _,i=max(c); x=ch.pop(i); ch[i:i]=kids(x[2][1])
        tr[k]=ch; todo+=[x[2][1] for x in ch if x[2][0]=='n']
    return tr
import time
DEEP=0
pos_of={int(k):i for i,k in enumerate(leaves)}
eye=np.array((-5.75,0.5,-16.8)); tgt=np.array((-5.75,0.5,-17.0)); fwd=tgt-eye; fwd/=np.linalg.norm(fwd)
right=np.cross(fwd,[0,1,0]); right/=np.linalg.norm(right); up=np.cross(right,fwd); tan=math.tan(math.pi/8)
def tri(i,o,d,lim):  # i index into leaves arrays
    pvec=np.cross(d,E2[i]); det=float(E1[i]@pvec)
    if abs(det)<1.19e-7: return None
    inv=1/det; tv=o-P0[i]; u=float(tv@pvec)*inv
    qv=np.cross(tv,E1[i]); v=float(d@qv)*inv; t=float(E2[i]@qv)*inv
    if u<0 or u>1 or v<0 or u+v>1 or t<=0 or t>=lim: return None
    return t
def walk(tr,rootk,o,d,leaf_tris):
    global DEEP
    inv=1/d; best=F; stack=[]; cur=('n',rootk); nn=nl=0; found=None
    while True:
        if cur[0]=='n':
            nn+=1; hits=[]
            for lo_,hi_,c in tr[cur[1]]:
                t1=(lo_-o)*inv; t2=(hi_-o)*inv
                a=max(0.0,float(np.minimum(t1,t2).max())); b=float(np.maximum(t1,t2).min())
                if a<=b and a<best: hits.append((a,c))
            hits.sort(key=lambda x:x[0])
            if hits:
                stack+= [x[1] for x in reversed(hits[1:])]; DEEP=max(DEEP,len(stack)); cur=hits[0][1]; continue
        else:
            for i in leaf_tris(cur[1]):
                nl+=1; t=tri(i,o,d,best)
                if t is not None: best=t; found=t
        if not stack: break
        cur=stack.pop()
    return found,nn,nl
trees={'implicit': (implicit_tree, implicit_root)}
t=time.time(); bt,r=karras(m[order]); trees['karras']=(collapse(bt,r),r); print("karras built",time.time()-t)
for rad in (8,):
    t=time.time(); bt,r=ploc(rad); trees['ploc%d'%rad]=(collapse(bt,r),r); print("ploc built",time.time()-t)
rng=np.random.default_rng(1); W,H=48,32
st={k:[0,0,0,0] for k in list(trees)+['sah']}; cnt=[0,0]; deep={}
leaf1=lambda i:[int(order[i])]
leafR=lambda c:[int(order[i]) for i in range(c*R,min((c+1)*R,N))]
for y in range(H):
  for x in range(W):
    px=((x+.5)/W*2-1)*tan*(W/H); py=(1-(y+.5)/H*2)*tan
    d=fwd+px*right+py*up; d/=np.linalg.norm(d); d[np.abs(d)<1e-9]=1e-9
    a=walk(sah,0,eye,d,lambda k:[pos_of[j] for j in sah_leaf(k)]); st['sah'][0]+=a[1]; st['sah'][1]+=a[2]; cnt[0]+=1
    for k,(tr,rt) in trees.items():
        b=walk(tr,rt,eye,d,leafR if k=='implicit' else leaf1); st[k][0]+=b[1]; st[k][1]+=b[2]
        assert (a[0] is None)==(b[0] is None)
    if a[0] is None: continue
    p=eye+d*a[0]; n=-d; r=rng.normal(size=3); r/=np.linalg.norm(r)
    if r@n<0: r=-r
    r[np.abs(r)<1e-9]=1e-9; o2=p+n*1e-3
    a2=walk(sah,0,o2,r,lambda k:[pos_of[j] for j in sah_leaf(k)]); st['sah'][2]+=a2[1]; st['sah'][3]+=a2[2]; cnt[1]+=1
    for k,(tr,rt) in trees.items():
        DEEP=0; b2=walk(tr,rt,o2,r,leafR if k=='implicit' else leaf1); st[k][2]+=b2[1]; st[k][3]+=b2[2]; deep[k]=max(deep.get(k,0),DEEP)
for k,v in st.items(): print(k,"primary: %.1f node + %.1f tri steps | bounce: %.1f node + %.1f tri steps"%(v[0]/cnt[0],v[1]/cnt[0],v[2]/cnt[1],v[3]/cnt[1]))

print('deepest stack (bounce rays):',deep)
