import os,sys
sys.path.insert(0,'.')
os.environ['ORBX_DEBUG']='1'
import numpy as np
import __graft_entry__ as g
pkg=g.load_package()
ex=pkg.ORBextractor(1000,1.2,8,20,7,max_width=640,max_height=480,max_batch=1024)
imgs=np.stack([pkg.synth_frame(i,640,480) for i in range(1024)])
ex.extract_batch(imgs,(0,0))
