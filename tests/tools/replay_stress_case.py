import sys, numpy as np, importlib
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package(); orc=g.load_oracle()
rng=np.random.default_rng(7)
# replay the random stream of tests/tools/stress_parity.py up to case 11
for case in range(12):
    w=int(rng.integers(120,1400)); h=int(rng.integers(100,1000)); nlev=int(rng.integers(1,9)); scale=float(rng.choice([1.1,1.2,1.25,1.5,2.0]))
    while min(w,h)/scale**(nlev-1)<70: nlev-=1
    if (w-32)/(h-32)<0.55: h=int(w*1.5)
    nf=int(rng.integers(50,6000)); ini=int(rng.integers(5,40)); mn=int(rng.integers(0,ini+1)); kind=rng.integers(0,4)
    if kind==0: img=pkg.synth_frame(int(rng.integers(0,1<<30)),w,h)
    elif kind==1: img=rng.integers(0,256,(h,w),dtype=np.uint8)
    elif kind==2: img=(128+rng.integers(-9,10,(h,w))).astype(np.uint8)
    else: img=np.kron(rng.integers(0,256,(h//16+1,w//16+1),dtype=np.uint8),np.ones((16,16),np.uint8))[:h,:w].copy()
    lap=(int(rng.integers(-10,w)),int(rng.integers(0,2*w)))
print(w,h,nf,nlev,scale,ini,mn,kind,lap)
ex=pkg.ORBextractor(nf,scale,nlev,ini,mn,max_width=w,max_height=h)
mono,kps,desc=ex(img,None,lap)
ref=orc.Oracle(nf,scale,nlev,ini,mn); rmono,rk,rd=ref.extract(img,lap)
print("n",len(kps),len(rk),"mono",mono,rmono, "quotas", ref.tables()['features_per_level'][:nlev])
for l in range(nlev):
    print("level",l,ex.level_size(l), "pyr",np.array_equal(ex.pyramid_level(0,l),ref.level(l)),"cand",np.array_equal(ex.debug_candidates(0,l),ref.candidates(l)),len(ref.candidates(l)),
          "blur",np.array_equal(ex.pyramid_level(0,l,True),ref.level(l,True)))
    s=ex.debug_selected(0,l); rkl=ref.level_keypoints(l)
    rs=((rkl["x"].astype(np.int64)-16)|((rkl["y"].astype(np.int64)-16)<<12)|(rkl["response"].astype(np.int64)<<24)).astype(np.uint32)
    print("   sel equal",np.array_equal(s,rs),len(s),len(rs), "same set", set(s.tolist())==set(rs.tolist()))
