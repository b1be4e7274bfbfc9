# kernel trace of the fused training-step pattern (render_and_semantic + render_class_distortions) at the C3 size: which kernels the 12 ms are
# usage: gpurun -- 'bash tools/trace_train_step.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/trace_train; rm -rf $D; mkdir -p $D; cd $R
cat > /tmp/ts.py <<'P'
import sys, torch
sys.path.insert(0, ".")
from streetunveiler_amd.gaussian_renderer import SurfelModel
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians
from streetunveiler_amd.train_pattern import fused_pattern, make_weights
P, W, H, dev = 3_000_000, 1920, 1080, "cuda:0"
cam = synthetic_camera(W, H).to(dev)
g = {k: v.to(dev).requires_grad_() for k, v in synthetic_gaussians(P, W, H).items()}
sem = torch.randint(0, 6, (P,), generator=torch.Generator().manual_seed(0)).to(dev); sem[sem == 4] = 2
pc = SurfelModel(g["means3D"], g["scales"], g["rotations"], g["opacities"], g["shs"], sem, 3, 3)
w = make_weights(H, W, dev); bg = torch.zeros(3, device=dev)
for _ in range(6):
    for t in g.values(): t.grad = None
    fused_pattern(cam, pc, bg, w)["loss"].backward()
torch.cuda.synchronize()
P
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- python /tmp/ts.py > $D/log.txt 2>&1
python tools/kstats.py $(find $D -name "*kernel_stats.csv" | head -1) | sort -k4 -n -r | head -40
