"""Would packing two instances' HALF-strips (8x4 pixels, 32 lanes each) into one wave-wide evaluation pay?  (EXPERIMENTS.md R5.9)

CPU estimate on the headline frame through the oracle (test infrastructure): for 500 sampled tiles and each 8x8 strip, the instances the
walk visits (list positions below the strip's largest n_contrib) are tested with the exact ellipse-vs-box reach test of wg_alpha.h against
the whole strip, its top / bottom and left / right halves and its four 4x4 quarters.  A wave that evaluated one instance per half (two
in-order queues per strip) would issue max(|top|, |bottom|) evaluations instead of |top U bottom|; printed as a ratio, with and without the
64-instance staging batches as synchronisation points.

    python tests/tools/half_strip_estimate.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))
import wg_scenes as S
from oracle import oracle as O
W,H,P=1920,1080,1000000
cloud=S.make_cloud(P,W,H,sh_degree=None,seed=0); cam=S.make_camera(W,H)
out=O.run_scene(cloud,cam,sh_degree=0); ctx=out["ctx"]
m2=ctx.get("means2D").astype(np.float64); co=ctx.get("conic_opacity").astype(np.float64)
pl=ctx.get("point_list"); ranges=ctx.get("ranges"); ncontrib=ctx.get("n_contrib")
gx,gy=(W+15)//16,(H+15)//16
rng=np.random.default_rng(0); sample=rng.choice(gx*gy,size=500,replace=False)
def reach(mx,my,A,B,Cc,tau2,vis,qx0,qy0,qx1,qy1):
    X0,X1,Y0,Y1=qx0-mx,qx1-mx,qy0-my,qy1-my
    inside=(X0<=0)&(X1>=0)&(Y0<=0)&(Y1>=0)
    best=np.full(len(mx),np.inf)
    for X in (X0,X1):
        dy=np.clip(-B*X/Cc,Y0,Y1); best=np.minimum(best,A*X*X+2*B*X*dy+Cc*dy*dy)
    for Y in (Y0,Y1):
        dx=np.clip(-B*Y/A,X0,X1); best=np.minimum(best,A*dx*dx+2*B*dx*Y+Cc*Y*Y)
    best=np.where(inside,0.0,best)
    return vis&(best<=tau2)
tot=dict(U=0,T=0,B=0,opsmax=0,opsmax_batch=0,L=0,R=0,opsLR=0,q=0,opsq=0)
for t in sample:
    tx,ty=t%gx,t//gx; x0,y0=tx*16,ty*16
    nc=ncontrib[y0:y0+16,x0:x0+16]
    lo,hi=ranges[t]
    for s in range(4):
        qx0,qy0=x0+8*(s&1),y0+8*(s>>1)
        if qx0>=W or qy0>=H: continue
        last=int(ncontrib[qy0:qy0+8,qx0:qx0+8].max())
        if last==0: continue
        ids=pl[lo:lo+last]
        mx,my=m2[ids,0],m2[ids,1]; A,B,Cc,o=co[ids,0],co[ids,1],co[ids,2],co[ids,3]
        tau2=2.0*np.log(np.maximum(255.0*o,1e-30)); vis=255.0*o>=1.0
        full=reach(mx,my,A,B,Cc,tau2,vis,qx0,qy0,min(qx0+7,W-1),min(qy0+7,H-1))
        top=reach(mx,my,A,B,Cc,tau2,vis,qx0,qy0,min(qx0+7,W-1),min(qy0+3,H-1))
        bot=reach(mx,my,A,B,Cc,tau2,vis,qx0,qy0+4,min(qx0+7,W-1),min(qy0+7,H-1)) if qy0+4<H else np.zeros(len(ids),bool)
        lef=reach(mx,my,A,B,Cc,tau2,vis,qx0,qy0,min(qx0+3,W-1),min(qy0+7,H-1))
        rig=reach(mx,my,A,B,Cc,tau2,vis,qx0+4,qy0,min(qx0+7,W-1),min(qy0+7,H-1))
        qs=[reach(mx,my,A,B,Cc,tau2,vis,qx0+4*(k&1),qy0+4*(k>>1),min(qx0+4*(k&1)+3,W-1),min(qy0+4*(k>>1)+3,H-1)) for k in range(4)]
        tot["U"]+=int(full.sum()); tot["T"]+=int(top.sum()); tot["B"]+=int(bot.sum())
        tot["opsmax"]+=max(int(top.sum()),int(bot.sum()))
        tot["L"]+=int(lef.sum()); tot["R"]+=int(rig.sum()); tot["opsLR"]+=max(int(lef.sum()),int(rig.sum()))
        tot["q"]+=sum(int(q.sum()) for q in qs); tot["opsq"]+=max(int(q.sum()) for q in qs)
        # batch-synchronised: the kernel stages 64 instances at a time (from the back in K9, from the front in K8)
        nb=(last+63)//64
        for b in range(nb):
            sl=slice(64*b,min(64*b+64,last))
            tot["opsmax_batch"]+=max(int(top[sl].sum()),int(bot[sl].sum()))
print(tot)
U=tot["U"]
print("ops per strip-eval now: 1.0; top/bottom halves: ideal %.3f, batch-synchronised %.3f; left/right ideal %.3f; quarters (16 lanes) ideal %.3f"%(tot["opsmax"]/U,tot["opsmax_batch"]/U,tot["opsLR"]/U,tot["opsq"]/U))
print("halves reached per reached strip: %.3f; quarters per reached strip %.3f"%((tot["T"]+tot["B"])/U, tot["q"]/U))
