import torch
p=torch.cuda.get_device_properties(0)
print(p)
for a in dir(p):
    if 'shared' in a.lower() or 'lds' in a.lower(): print(a, getattr(p,a))
