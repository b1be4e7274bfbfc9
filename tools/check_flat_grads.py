import sys, math, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from streetunveiler_amd.synthetic import *
from streetunveiler_amd.parallel import _storage_groups
P,W,H,dev=20000,320,200,'cuda:0'
cam=synthetic_camera(W,H); g={k:v.to(dev).requires_grad_() for k,v in synthetic_gaussians(P,W,H,scale_lo=2e-3,scale_hi=3e-2).items()}
s=GaussianRasterizationSettings(H,W,math.tan(cam.FoVx/2),math.tan(cam.FoVy/2),torch.zeros(3,device=dev),1.0,cam.world_view_transform.to(dev),cam.full_proj_transform.to(dev),3,cam.camera_center.to(dev),False,False)
m2d=torch.zeros(P,3,device=dev,requires_grad=True)
c,r,a=GaussianRasterizer(s)(means3D=g['means3D'],means2D=m2d,shs=g['shs'],opacities=g['opacities'],scales=g['scales'],rotations=g['rotations'])
(c.sum()+a.sum()).backward()
gr=[g[k].grad for k in ['means3D','shs','opacities','scales','rotations']]
print('storages', len({x.untyped_storage().data_ptr() for x in gr}))
groups,rest=_storage_groups(gr); print('flat groups', len(groups), [x.numel() for x in groups], 'rest', len(rest), 'expect numel', P*58)
