import sys, torch, numpy as np
sys.path.insert(0,'.')
from eda_amd import ext
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it*1e3
rng=np.random.default_rng(0)
for n,m in [(512,256),(513,256),(400,256),(256,128),(1024,256),(2048,256)]:
    x=torch.from_numpy(rng.uniform(1,3,(8,n,3)).astype(np.float32)).cuda()
    us=t(lambda: ext.furthest_point_sampling(x,m))
    print(f"uniform n={n} m={m}: {us:.1f} us = {us/(m-1):.2f} us/round")
# FPS-ordered input like SA4
x=torch.from_numpy(rng.uniform(1,3,(8,4096,3)).astype(np.float32)).cuda()
i=ext.furthest_point_sampling(x,512)
y=torch.gather(x,1,i.long()[...,None].expand(-1,-1,3)).contiguous()
us=t(lambda: ext.furthest_point_sampling(y,256)); print(f"fps-ordered n=512 m=256: {us:.1f} us = {us/255:.2f} us/round")
