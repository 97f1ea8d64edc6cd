import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
import fisheye_synth as fs
import test_fisheye_gpu as T
kl, dl, kr, dr, R12, t12 = T._scenario(pkg, 400, 150, 90, seed=550)
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7); sigma2 = ex.GetScaleSigmaSquares(); ex.close()
want = orc.fisheye_stereo_matches(kl, dl, 150, kr, dr, 90, fs.CAM1, fs.CAM2, R12, t12, sigma2)
m = pkg.ORBmatcher(4096, 4096)
got = m.ComputeStereoFishEyeMatches(kl, dl, 150, kr, dr, 90, fs.CAM1, fs.CAM2, R12, t12, sigma2)
d = np.nonzero(got[3].view(np.uint32) != want[3].view(np.uint32))[0]
print("n differing", len(d), "of", (want[3] > 0).sum())
for i in d[:10]:
    print(i, got[3][i], want[3][i], int(got[3].view(np.uint32)[i]) - int(want[3].view(np.uint32)[i]), got[4][i], want[4][i])
