"""-m gpu: the drop-in boundary's contract on streams, threads, layouts and repeated backwards ("kernels on the current
torch stream", "per-call state only", inputs borrowed and never mutated).  The backward is atomic-free, so "same result" below always
means bit for bit."""
import threading

import pytest
import torch

from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads
from tests.gpu_util import DEV, settings_for
from diff_surfel_rasterization import GaussianRasterizer

pytestmark = pytest.mark.gpu

NAMES = ("means3D", "opacities", "scales", "rotations", "shs")


def _scene(P, W, H, seed):
    cam = synthetic_camera(W, H)
    g = synthetic_gaussians(P, W, H, seed=seed)
    dc, da = synthetic_upstream_grads(W, H, seed=seed + 1)
    return cam, g, dc, da


def _step(cam, g, dc, da, deg=3, leaves=None, retain=False):
    """One forward + backward through the operator on whatever stream is current; everything returned stays on the device."""
    t = leaves or {k: g[k].to(DEV).requires_grad_() for k in NAMES}
    m2 = torch.zeros(t["means3D"].shape[0], 3, device=DEV, requires_grad=True)
    color, radii, allmap = GaussianRasterizer(settings_for(cam, [0.1, 0.2, 0.3], deg))(
        means3D=t["means3D"], means2D=m2, opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], shs=t["shs"])
    loss = (color * dc).sum() + (allmap * da).sum()
    grads = torch.autograd.grad(loss, [t[k] for k in NAMES] + [m2], retain_graph=retain)
    return dict(color=color.detach(), radii=radii, allmap=allmap.detach(), **{"d" + k: v for k, v in zip(NAMES + ("means2D",), grads)}), loss


def _same(a, b, tag):
    for k in a:
        assert torch.equal(a[k], b[k]), f"{tag}: {k} differs (max |d| = {(a[k].double() - b[k].double()).abs().max().item():.3e})"


def test_side_stream_gives_the_default_streams_bits():
    """The operator queues everything on torch's CURRENT stream (forward read-back included): under `torch.cuda.stream(side)` with the
    default stream kept busy by unrelated work, the outputs and gradients are the default-stream run's, bit for bit."""
    cam, g, dc, da = _scene(40_000, 640, 360, 11)
    dc, da = dc.to(DEV), da.to(DEV)
    base, _ = _step(cam, g, dc, da)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    busy = torch.randn(4096, 4096, device=DEV)
    for _ in range(3):
        busy = busy @ busy * 1e-3                      # default stream: long-running, unrelated
        side.wait_stream(torch.cuda.current_stream())  # (the inputs were produced on the default stream)
        with torch.cuda.stream(side):
            got, _ = _step(cam, g, dc, da)
        side.synchronize()
        _same(base, got, "side stream")
    torch.cuda.synchronize()


def test_two_threads_on_two_streams():
    """Per-call state only: two python threads, each with its own stream and its own scene, interleave forward and backward calls (the
    forward's read-back blocks only its own thread); every iteration of either thread reproduces that scene's single-threaded bits."""
    scenes = [_scene(30_000, 512, 288, 21), _scene(50_000, 640, 360, 31)]
    scenes = [(cam, g, dc.to(DEV), da.to(DEV)) for cam, g, dc, da in scenes]
    base = [_step(*s)[0] for s in scenes]
    torch.cuda.synchronize()
    errors = []

    def worker(i):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for it in range(6):
                    got, _ = _step(*scenes[i])
                    stream.synchronize()
                    _same(base[i], got, f"thread {i} iteration {it}")
        except Exception as e:                          # noqa: BLE001 -- reported by the main thread
            errors.append(f"thread {i}: {type(e).__name__}: {e}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads: t.start()
    for t in threads: t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a worker thread hung"
    assert not errors, "\n".join(errors)


def test_strided_and_double_inputs_are_borrowed_not_required_contiguous():
    """The reference's callers hand over whatever the model's getters return; the shim makes float32 contiguous copies when (and only
    when) it has to, never writes to an input, and the gradients come back in the INPUT's layout and dtype."""
    cam, g, dc, da = _scene(20_000, 480, 272, 41)
    dc, da = dc.to(DEV), da.to(DEV)
    base, _ = _step(cam, g, dc, da)
    P = g["means3D"].shape[0]
    # transposed storage for the [P, 3] / [P, 4] / [P, 2] tensors, a strided slice of a wider buffer for the SH rows, float64 opacities
    leaves = {}
    for k in ("means3D", "scales", "rotations"):
        leaves[k] = g[k].t().contiguous().to(DEV).t().requires_grad_()
        assert not leaves[k].is_contiguous()
    wide = torch.zeros(P, 20, 3, device=DEV); wide[:, 2:18] = g["shs"].to(DEV)
    leaves["shs"] = wide[:, 2:18].detach().requires_grad_()
    assert not leaves["shs"].is_contiguous()
    leaves["opacities"] = g["opacities"].double().to(DEV).requires_grad_()
    before = {k: v.detach().clone() for k, v in leaves.items()}
    got, _ = _step(cam, g, dc, da, leaves=leaves)
    for k, v in leaves.items():
        assert torch.equal(v.detach(), before[k]), f"input {k} was written to"
    assert got["dopacities"].dtype == torch.float64 and got["dmeans3D"].shape == (P, 3) and got["dshs"].shape == (P, 16, 3)
    got["dopacities"] = got["dopacities"].float()
    _same(base, got, "strided / float64 inputs")


def test_backward_twice_with_retain_graph_and_once_without():
    """The saved state buffers are read-only for the backward (its records live in a per-call workspace): with `retain_graph=True` a
    second backward returns the first one's bits; without it torch's own "backward through the graph a second time" error is raised,
    as for any autograd.Function."""
    cam, g, dc, da = _scene(20_000, 480, 272, 51)
    dc, da = dc.to(DEV), da.to(DEV)
    t = {k: g[k].to(DEV).requires_grad_() for k in NAMES}
    m2 = torch.zeros(t["means3D"].shape[0], 3, device=DEV, requires_grad=True)
    color, radii, allmap = GaussianRasterizer(settings_for(cam, [0., 0., 0.], 3))(
        means3D=t["means3D"], means2D=m2, opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], shs=t["shs"])
    loss = (color * dc).sum() + (allmap * da).sum()
    leaves = [t[k] for k in NAMES] + [m2]
    first = torch.autograd.grad(loss, leaves, retain_graph=True)
    second = torch.autograd.grad(loss, leaves, retain_graph=False)
    for k, a, b in zip(NAMES + ("means2D",), first, second):
        assert torch.equal(a, b), f"second backward: d{k} differs"
    with pytest.raises(RuntimeError, match="second time|already been freed"):
        torch.autograd.grad(loss, leaves)
    # ... and .backward() ACCUMULATES into .grad as for any other op: two frames' gradients add up
    for _ in range(2):
        c, _, a = GaussianRasterizer(settings_for(cam, [0., 0., 0.], 3))(
            means3D=t["means3D"], means2D=m2, opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], shs=t["shs"])
        ((c * dc).sum() + (a * da).sum()).backward()
    for k, a in zip(NAMES + ("means2D",), first):
        leaf = m2 if k == "means2D" else t[k]
        assert torch.equal(leaf.grad, a + a), f"accumulated d{k} is not twice one frame's"


def test_backward_captures_into_a_hip_graph():
    """The backward has no host synchronisation, no allocation by the library and no process-wide state: `sr_backward` (memset of the
    written flags, K7, K8) is captured into a HIP graph and every replay returns the eager call's bits.  (The forward cannot be: its
    `num_rendered` read-back is the reference's own host sync.  Captured through `_C` directly: `torch.autograd.grad` inside
    `torch.cuda.graph` dumps core on this torch / ROCm pair for pure-torch graphs too.)"""
    from diff_surfel_rasterization import _C
    cam, g, dc, da = _scene(10_000, 256, 256, 61)
    g = {k: v.to(DEV) for k, v in g.items()}
    dc, da = dc.to(DEV), da.to(DEV)
    s = settings_for(cam, [0., 0., 0.], 3)
    e = torch.empty(0, device=DEV)
    fwd = lambda: _C.rasterize_gaussians(s.bg, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, s.viewmatrix, s.projmatrix,
                                         s.tanfovx, s.tanfovy, 256, 256, g["shs"], 3, s.campos, False, False)
    D, color, allmap, radii, geom, binning, img = fwd()[:7]
    bwd = lambda: _C.rasterize_gaussians_backward(s.bg, g["means3D"], radii, e, g["scales"], g["rotations"], 1.0, e, s.viewmatrix, s.projmatrix,
                                                  s.tanfovx, s.tanfovy, dc, da, g["shs"], 3, s.campos, geom, D, binning, img, False)
    ref = bwd()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        bwd()                                         # warm-up on a side stream, as torch's capture rules ask
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = bwd()
    for _ in range(3):
        for o in out: o.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        for k, (a, b) in enumerate(zip(ref, out)):
            assert torch.equal(a, b), f"graph replay: gradient {k} differs from the eager call"


def test_binning_capacity_forward_without_the_host_readback():
    """SR_FLAG_BINNING_CAPACITY (round 6): the operator without the `num_rendered` read-back.  A capacity that holds the frame gives the
    default mode's bits -- images, radii, every gradient -- and the device status [D, visible, 0]; a capacity that does not gives the
    background, zero gradients, status[2] = 1 with the exact D in status[0], and touches nothing beyond its buffers."""
    cam, g, dc, da = _scene(60_000, 640, 360, 71)
    dc, da = dc.to(DEV), da.to(DEV)
    bg = [0.1, 0.2, 0.3]

    def step(**kw):
        t = {k: g[k].to(DEV).requires_grad_() for k in NAMES}
        m2 = torch.zeros(t["means3D"].shape[0], 3, device=DEV, requires_grad=True)
        r = GaussianRasterizer(settings_for(cam, bg, 3), **kw)
        color, radii, allmap = r(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], shs=t["shs"])
        grads = torch.autograd.grad((color * dc).sum() + (allmap * da).sum(), [t[k] for k in NAMES] + [m2])
        return dict(color=color.detach(), radii=radii, allmap=allmap.detach(), **{"d" + k: v for k, v in zip(NAMES + ("means2D",), grads)}), r

    base, _ = step()
    from tests.gpu_util import run_hip_raw
    D = int(run_hip_raw(g, cam, bg, 3)["D"])
    assert D > 100_000
    got, r = step(binning_capacity=int(1.25 * D))
    assert r.last_status.tolist()[0] == D and r.last_status.tolist()[2] == 0 and r.last_status.tolist()[1] == int((base["radii"] > 0).sum())
    _same(base, got, "capacity 1.25 D")
    got, r = step(binning_capacity=D)                 # exactly full
    assert r.last_status.tolist() == [D, int((base["radii"] > 0).sum()), 0]
    _same(base, got, "capacity D")
    got, r = step(binning_capacity=D // 2)            # does not fit
    st = r.last_status.tolist()
    assert st[0] == D and st[2] == 1 and st[1] == 0, st
    assert torch.equal(got["radii"], base["radii"])   # (K1 ran: the radii are the frame's)
    want = torch.tensor(bg, device=DEV).view(3, 1, 1).expand_as(got["color"])
    assert torch.equal(got["color"], want) and not got["allmap"].any(), "an overflowed frame must hold the background"
    for k in got:
        if k.startswith("d"):
            assert not got[k].any(), f"an overflowed frame must have zero gradients ({k})"
    # under no_grad (forward-only state) the same
    with torch.no_grad():
        t = {k: g[k].to(DEV) for k in NAMES}
        r = GaussianRasterizer(settings_for(cam, bg, 3), binning_capacity=int(1.1 * D))
        color, radii, allmap = r(means3D=t["means3D"], means2D=torch.zeros(t["means3D"].shape[0], 3, device=DEV), opacities=t["opacities"], scales=t["scales"],
                                 rotations=t["rotations"], shs=t["shs"])
    assert torch.equal(color, base["color"]) and torch.equal(allmap, base["allmap"]) and r.last_status.tolist()[2] == 0


def test_whole_step_captures_into_a_hip_graph_with_a_binning_capacity():
    """With a binning capacity the forward has no host wait either: forward AND backward are captured into ONE HIP graph (through `_C`, as
    in the backward-only test above), and every replay -- also with new values in the same input tensors -- returns the eager bits."""
    from diff_surfel_rasterization import _C
    cam, g, dc, da = _scene(10_000, 256, 256, 61)
    g = {k: v.to(DEV) for k, v in g.items()}
    dc, da = dc.to(DEV), da.to(DEV)
    s = settings_for(cam, [0., 0., 0.], 3)
    e = torch.empty(0, device=DEV)
    D0 = _C.rasterize_gaussians(s.bg, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, s.viewmatrix, s.projmatrix,
                                s.tanfovx, s.tanfovy, 256, 256, g["shs"], 3, s.campos, False, False)[0]
    cap = int(1.5 * D0)

    def whole(capacity):
        kw = {} if capacity is None else dict(binning_capacity=capacity)
        D, color, allmap, radii, geom, binning, img = _C.rasterize_gaussians(s.bg, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, s.viewmatrix,
                                                                             s.projmatrix, s.tanfovx, s.tanfovy, 256, 256, g["shs"], 3, s.campos, False, False, **kw)[:7]
        grads = _C.rasterize_gaussians_backward(s.bg, g["means3D"], radii, e, g["scales"], g["rotations"], 1.0, e, s.viewmatrix, s.projmatrix,
                                                s.tanfovx, s.tanfovy, dc, da, g["shs"], 3, s.campos, geom, D, binning, img, False, **kw)
        return [color, allmap, radii] + [x for x in grads if x is not None and x.numel()], (_C.forward_status(geom, 10_000) if capacity is not None else None)

    ref, _ = whole(None)
    ref = [x.clone() for x in ref]
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        whole(cap)                                    # warm-up on a side stream (rank self-check, allocator)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out, status = whole(cap)
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        assert status.tolist() == [D0, int((ref[2] > 0).sum()), 0]
        for a, b in zip(ref, out):
            assert torch.equal(a, b), "graph replay differs from the eager step"
    # new parameter values in the SAME tensors: the replay renders the new scene
    with torch.no_grad():
        g["opacities"].mul_(0.5); g["means3D"][:, 0].add_(0.05)
    ref2, _ = whole(None)
    ref2 = [x.clone() for x in ref2]
    graph.replay(); torch.cuda.synchronize()
    assert status.tolist()[2] == 0
    for k, (a, b) in enumerate(zip(ref2, out)):
        assert torch.equal(a, b), f"graph replay on updated inputs: output {k} differs from the eager step ({int((a != b).sum())} of {a.numel()} elements, max |d| {(a.double() - b.double()).abs().max().item():.3e})"
