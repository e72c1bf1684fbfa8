import sys, time
sys.path.insert(0,'/root/repo')
import torch
from strolle_amd import Engine, scenes, CameraMode
e=Engine(0); scenes.build_cornell(e); e.set_seed(0)
size=(1920,1080)
d=scenes.cornell_camera(size, CameraMode.IMAGE); c=e.create_camera(d)
out=torch.zeros((size[1],size[0],4),device='cuda'); st=torch.cuda.current_stream().cuda_stream
def step():
    e.update_camera(c,d); e.tick(st); e.render_camera(c,out.data_ptr(),st)
for _ in range(12): step()
torch.cuda.synchronize()
# enqueue-only cost: tiny frame so the GPU is never the limit
e2=Engine(0); scenes.build_cornell(e2); d2=scenes.cornell_camera((64,64), CameraMode.IMAGE); c2=e2.create_camera(d2)
out2=torch.zeros((64,64,4),device='cuda')
for _ in range(12):
    e2.update_camera(c2,d2); e2.tick(st); e2.render_camera(c2,out2.data_ptr(),st)
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(200):
    e2.update_camera(c2,d2); e2.tick(st); e2.render_camera(c2,out2.data_ptr(),st)
t1=time.perf_counter()-t
torch.cuda.synchronize()
t2=time.perf_counter()-t
print("64x64: host enqueue per frame %.3f ms, incl. drain %.3f ms"%(t1/200*1e3, t2/200*1e3))
t=time.perf_counter()
for _ in range(100): step()
t1=time.perf_counter()-t
torch.cuda.synchronize()
t2=time.perf_counter()-t
print("1080p: host loop per frame %.3f ms, total per frame %.3f ms"%(t1/100*1e3, t2/100*1e3))
