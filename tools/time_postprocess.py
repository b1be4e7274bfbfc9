import sys, time, torch, math
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streetunveiler_amd.gaussian_renderer import postprocess_allmap, PipelineParams
from streetunveiler_amd.synthetic import synthetic_camera
W,H=1920,1080; dev='cuda:0'
cam=synthetic_camera(W,H).to(dev)
allmap=(torch.rand(7,H,W,device=dev)+0.1).requires_grad_()
pipe=PipelineParams()
def step():
    allmap.grad=None
    o=postprocess_allmap(cam,pipe,allmap)
    normal_error = (1 - (o["rend_normal"] * o["surf_normal"]).sum(dim=0))[None]
    loss = 0.05*normal_error.mean() + 100*o["rend_dist"].mean() + 0.1*o["rend_alpha"].mean()
    loss.backward()
for _ in range(5): step()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print('postprocess+loss fwd+bwd ms', (time.perf_counter()-t)/20*1e3)
