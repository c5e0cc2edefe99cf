import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT,"wild-gaussians_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import numpy as np, torch
import wg_scenes as S
from oracle import oracle
from wg_testlib import run_hip
W,H,P=64,64,int(sys.argv[1]) if len(sys.argv)>1 else 50
cam=S.make_camera(W,H)
cloud=S.make_cloud(P,W,H,sh_degree=None,seed=3,scale_mult=float(sys.argv[2]) if len(sys.argv)>2 else 20.0)
cot=S.make_cotangent(W,H)
o=oracle.run_scene(cloud,cam,cotangent=cot)
h=run_hip(cloud,cam,sh_degree=0,cotangent=cot)
print("R",o["num_rendered"])
for k in ["colors_precomp","means2D","opacities"]:
    a=h["grads"][k]; r=o["grads"][k].reshape(a.shape)
    den=np.abs(r).max(axis=0)+1e-20
    print(k,"col err",np.abs(a-r).max(axis=0)/den, "ratio of sums", a.sum(axis=0)/ (r.sum(axis=0)+1e-30))
    bad=np.argwhere(np.abs(a-r)>1e-3*den)
    print("  n bad",len(bad), bad[:5].tolist())
    for i,c in bad[:5]:
        print("   ",i,c,a[i,c],r[i,c])
