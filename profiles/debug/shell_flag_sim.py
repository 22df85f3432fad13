import numpy as np, sys
sys.path.insert(0,'/root/repo')
from ascendpathtracing_amd import gen_data
ns=10000
scene=gen_data.gen_scene(ns, seed=1)
tab=scene[:10*ns].reshape(10,ns).astype(np.float64)
r2,cx,cy,cz=tab[0],tab[1],tab[2],tab[3]
g=gen_data.build_grid(scene, ns)
n=g[2:5]; fl=g[13:26].view(np.float32).astype(np.float64)
gmin,gmax,cell=fl[0:3],fl[3:6],fl[6:9]; margin=fl[12]
off_cellslot=int(g[26]); n0,n1,n2=(int(x) for x in n)
cs=g[off_cellslot:off_cellslot+(n0+2)*(n1+2)*(n2+2)].reshape(n2+2,n1+2,n0+2)
rng=np.random.default_rng(0)
NP=1500
# camera rays
w,h=1920,1080
cam=np.array([50,52,295.6]); gd=np.array([0,-0.042612,-1.0]); gd/=np.linalg.norm(gd)
cxv=np.array([w*0.5135/h,0,0]); cyv=np.cross(cxv,gd); cyv=cyv/np.linalg.norm(cyv)*0.5135
u=rng.random((NP,2))-0.5
d=cxv[None]*u[:,0:1]+cyv[None]*u[:,1:2]+gd[None]
o=cam[None]+140*d
d/=np.linalg.norm(d,axis=1,keepdims=True)
stats={"segs":0,"flag_any":0,"pairs":0,"cells":0,"wallhit":0}
hist=np.zeros(16,int)
for depth in range(8):
    oc=np.stack([cx[None]-o[:,0:1],cy[None]-o[:,1:2],cz[None]-o[:,2:3]],axis=2)  # NP,ns,3
    b=(oc*d[:,None,:]).sum(2); c=(oc*oc).sum(2)-r2[None]
    disc=b*b-c
    q=np.sqrt(np.where(disc>=0,disc,np.nan))
    t0=b-q; t1=b+q
    t=np.where(t0>1e-4,t0,t1); t=np.where(t>1e-4,t,1e20); t=np.where(np.isnan(t),1e20,t)
    idx=t.argmin(1); tm=t[np.arange(NP),idx]
    for p in range(NP):
        # DDA
        e=o[p]-gmin; ci=np.floor(e/cell).astype(int)
        if (ci<0).any() or (ci>=n).any(): 
            hist[15]+=1; stats["segs"]+=1; stats["flag_any"]+=1; stats["pairs"]+=4; continue
        dd=d[p]; step=np.where(dd>0,1,-1)
        with np.errstate(divide='ignore'):
            tmax=np.where(dd!=0,((ci+(dd>0))*cell-e)/dd,np.inf); tdel=np.where(dd!=0,cell/np.abs(dd),np.inf)
        flags=0; ncell=0
        while True:
            ncell+=1
            flags|=int(cs[ci[2]+1,ci[1]+1,ci[0]+1])&15
            te=tmax.min()
            if tm[p] < te*0.999-margin: break
            a=int(tmax.argmin()); ci[a]+=step[a]; tmax[a]+=tdel[a]
            if ci[a]<0 or ci[a]>=n[a]: flags|=15; break
        hist[flags]+=1
        stats["segs"]+=1; stats["flag_any"]+= flags!=0; stats["pairs"]+=bin(flags).count("1"); stats["cells"]+=ncell
        stats["wallhit"]+= (idx[p]<6 or idx[p]==ns-1)
    hp=o+d*tm[:,None]
    nrm=hp-np.stack([cx[idx],cy[idx],cz[idx]],1); nrm/=np.linalg.norm(nrm,axis=1,keepdims=True)
    d=d-2*(d*nrm).sum(1,keepdims=True)*nrm; o=hp
print(stats, {k:v/stats["segs"] for k,v in stats.items()})
print("flag hist", hist)
