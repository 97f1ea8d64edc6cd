"""How often do the device's (ocml) and the host's (glibc) double-precision tan / atan2 / cos / sin differ AFTER narrowing to float?
(torch evaluates float64 tensors with the ocml double functions on the GPU.)"""
import numpy as np, torch
rng = np.random.default_rng(0)
x = rng.uniform(0, 1.55, 2_000_000).astype(np.float32).astype(np.float64)
xd = torch.from_numpy(x).cuda()
for name, fh, fd in (("tan", np.tan, torch.tan), ("cos", np.cos, torch.cos), ("sin", np.sin, torch.sin), ("atan", np.arctan, torch.atan)):
    h = fh(x); d = fd(xd).cpu().numpy()
    ulp = np.abs(h.view(np.int64) - d.view(np.int64))
    nf = (h.astype(np.float32) != d.astype(np.float32)).mean()
    print(f"{name}: double results differ in {np.mean(ulp > 0):.4f} of inputs, max {ulp.max()} ulp; after narrowing to float: {nf:.2e}")
y = rng.uniform(-3, 3, 2_000_000).astype(np.float32).astype(np.float64); z = rng.uniform(0.1, 6, 2_000_000).astype(np.float32).astype(np.float64)
h = np.arctan2(y, z); d = torch.atan2(torch.from_numpy(y).cuda(), torch.from_numpy(z).cuda()).cpu().numpy()
print("atan2: double differ", np.mean(h != d), "float differ", (h.astype(np.float32) != d.astype(np.float32)).mean())
