import sys, time, importlib, numpy as np
sys.path.insert(0,'.')
import torch
pkg=importlib.import_module("wut-cuda-orb-slam3_amd")
ex=pkg.ORBextractor(5000,1.2,8,20,7)
img1=pkg.synth_frame(7,640,480); img2=np.roll(np.roll(img1,-6,0),9,1)
_,k1,d1=ex(img1,None,(0,0)); k1,d1=k1.copy(),d1.copy()
_,k2,d2=ex(img2,None,(0,0)); k2,d2=k2.copy(),d2.copy()
prev=np.stack([k1['x'],k1['y']],1)
m=pkg.ORBmatcher(8192,8192)
m.SearchForInitialization(k1,d1,k2,d2,prev,(0,640,0,480),100,0.9,True)
t=time.time()
for _ in range(10): n,_,_=m.SearchForInitialization(k1,d1,k2,d2,prev,(0,640,0,480),100,0.9,True)
print("SearchForInitialization: %.2f ms per call, n1=%d (level 0: %d), n2=%d, matches=%d"%((time.time()-t)*100,len(k1),(k1['octave']==0).sum(),len(k2),n))
