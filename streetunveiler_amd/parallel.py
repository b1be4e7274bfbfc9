"""Frame-sharded data parallelism for the render operator.

The path shards by camera: every rank holds the full Gaussian set, renders its own frame (K1..K8 need no
collective), and the ranks meet once per optimisation step:
  * SUM all-reduce of the parameter gradients (58 floats / Gaussian with SH degree 3);
  * densification statistics, which are per-view norms and therefore cannot be derived from the summed
    gradient: SUM of xyz_gradient_accum / denom, MAX of max_radii2D
    [REF /root/reference/scene/gaussian_model.py:555-557; /root/reference/train.py:166-169].
  * the SH part of that gradient (192 of the 232 B per Gaussian) can be exchanged in factored form instead
    (`factored_sh_exchange`): the SH adjoint is linear in the clamp-masked colour gradient (12 B per Gaussian) and
    its only other per-view input is the camera position, so ranks ALL-GATHER the colour gradients and every rank
    expands sum_v basis(dir_v) (x) g_v locally (csrc/preprocess.hip sh_gradient_expand_kernel).  The result is the
    same all-reduced dL_dsh (float summation order aside) for 8x (2 ranks) .. 2.3x (8 ranks) fewer bytes over xGMI.
One process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI on ROCm); "gloo" for the CPU tests.
xGMI is point-to-point, so few large messages beat many small ones: small tensors are packed into one
bucket, large ones (the SH gradient is 83 % of the bytes) are reduced in place without a staging copy.
"""
from __future__ import annotations

import contextlib
import contextvars
import os
import weakref
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

_BUCKET_INPLACE_BYTES = 32 << 20  # tensors at least this large are all-reduced in place
# replicated camera lists (the tensor OBJECTS, weakly held, with the version they were checked at) already compared with the cameras actually rendered
_CAMERA_LISTS_CHECKED = {}   # id(tensor) -> (weak reference to it, version)


def _exchange_wanted(group=None) -> bool:
    """True when the frame-parallel collectives have to run: a group of more than one rank -- or, with
    SURFEL_EXCHANGE_SINGLE_RANK=1, any initialised group (a one-rank RCCL group on a single GPU runs the very same calls:
    tests/test_gpu_parity.py::test_factored_sh_exchange_over_rccl_with_one_rank)."""
    if not dist.is_initialized():
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("SURFEL_EXCHANGE_SINGLE_RANK") == "1"


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment. Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("SURFEL_EXCHANGE_SINGLE_RANK") == "1") and not dist.is_initialized():
        if backend is None:
            # SURFEL_DIST_BACKEND=gloo lets the N>1 path be exercised on a box with fewer GPUs than ranks
            backend = os.environ.get("SURFEL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            if local_rank >= torch.cuda.device_count():
                raise RuntimeError(f"LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPUs visible (one process per GPU)")
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def frames_for_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """Camera k -> rank k mod world."""
    return [k for k in range(n_frames) if k % world == rank]


def _storage_groups(grads: Sequence[torch.Tensor]):
    """Group tensors that are contiguous views of one storage and together cover one contiguous span of it."""
    by_storage = {}
    for g in grads:
        by_storage.setdefault(g.untyped_storage().data_ptr(), []).append(g)
    groups, singles = [], []
    for members in by_storage.values():
        members.sort(key=lambda t: t.storage_offset())
        covered = all(m.is_contiguous() for m in members) and all(
            members[i].storage_offset() + members[i].numel() == members[i + 1].storage_offset() for i in range(len(members) - 1))
        if len(members) > 1 and covered and len({m.dtype for m in members}) == 1:
            first = members[0]
            total = sum(m.numel() for m in members)
            groups.append(torch.empty(0, dtype=first.dtype, device=first.device).set_(first.untyped_storage(), first.storage_offset(), (total,)))
        else:
            singles.extend(members)
    return groups, singles


def allreduce_gradients(grads: Sequence[torch.Tensor], group=None, average: bool = False) -> None:
    """In-place SUM (or mean) all-reduce of a list of gradient tensors across the frame-parallel ranks.

    Gradients that the rasterizer's backward carved out of one flat buffer (diff_surfel_rasterization._C) are reduced
    with ONE collective over that buffer; anything else: large tensors in place, small ones through one bucket."""
    if not _exchange_wanted(group):
        return
    world = dist.get_world_size(group)
    grads = [g for g in grads if g is not None and g.numel() > 0]
    flat_groups, rest = _storage_groups(grads)
    small = [g for g in rest if g.numel() * g.element_size() < _BUCKET_INPLACE_BYTES]
    large = flat_groups + [g for g in rest if g.numel() * g.element_size() >= _BUCKET_INPLACE_BYTES]
    handles = []
    for g in sorted(large, key=lambda t: -t.numel()):  # largest first so the long transfer starts immediately
        handles.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True))
    with STALLS.waiting(bool(grads) and grads[0].is_cuda):
        if small:
            flat = torch.cat([g.reshape(-1) for g in small])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            off = 0
            for g in small:
                n = g.numel()
                g.copy_(flat[off:off + n].view_as(g))
                off += n
        for h in handles:
            h.wait()
    if average:
        for g in grads:
            g.div_(world)


def pack_visibility(visible: torch.Tensor) -> torch.Tensor:
    """[P] bool -> uint8 [ceil(P / 8)], bit i & 7 of byte i >> 3 = visible[i] (little-endian bits: numpy's packbits(bitorder="little"))."""
    v = visible.reshape(-1).to(torch.uint8)
    pad = (-v.numel()) % 8
    if pad:
        v = torch.cat([v, v.new_zeros(pad)])
    w = (1 << torch.arange(8, device=v.device, dtype=torch.int32)).to(torch.uint8)
    return (v.view(-1, 8) * w).sum(1, dtype=torch.int32).to(torch.uint8)


def unpack_visibility(bits: torch.Tensor, P: int) -> torch.Tensor:
    w = (1 << torch.arange(8, device=bits.device, dtype=torch.int32)).to(torch.uint8)
    return ((bits.view(-1, 1) & w) != 0).reshape(-1)[:P]


def union_of_visible_rows(visible: torch.Tensor, group=None) -> torch.Tensor:
    """[P] bool: the Gaussians this rank's frame(s) reached (radii > 0) -> [P] bool: reached on AT LEAST ONE rank.  One all-gather of the
    packed bitmaps: P / 8 bytes per rank (375 kB at 3 M Gaussians against 696 MB of gradients)."""
    bits = pack_visibility(visible)
    world = dist.get_world_size(group)
    rows = torch.empty(world * bits.numel(), dtype=torch.uint8, device=bits.device)
    dist.all_gather_into_tensor(rows, bits, group=group)
    acc = rows.view(world, -1)[0].clone()
    for r in range(1, world):
        acc |= rows.view(world, -1)[r]
    return unpack_visibility(acc, visible.numel())


# Bytes this process put on the wire per call of the exchange helpers below (diagnostic: bench.py --gpus N reports it per step)
WIRE = {"bytes_payload": 0, "bytes_dense_equivalent": 0, "rows_union": 0, "rows_total": 0, "calls": 0}


def allreduce_visible_rows(grads: Sequence[torch.Tensor], visible: torch.Tensor, group=None, dense_above: float = 0.85) -> dict:
    """The gradient exchange WITHOUT the zeros (round 6).  A rank's gradient rows are exact zeros for every Gaussian its frame did not
    reach, and on a street most Gaussians are behind the camera: the benchmark rig sees 86 %, a camera inside the scene 17 %
    (bench.py camera_inside_scene) -- 83 % of what the dense all-reduce moves over xGMI is zeros.  Here the ranks
      1. all-gather their packed visibility bitmaps (P / 8 bytes) and OR them: the rows reached on at least one rank;
      2. gather those rows of every gradient tensor into ONE [n_union, columns] buffer, SUM all-reduce it, scatter the sums back.
    Rows outside the union are zero on every rank and stay untouched.  `grads`: tensors with leading dimension P (views of the operator's
    flat gradient buffer are fine); `visible` [P] bool = radii > 0 of this rank's frame (OR over its frames with accumulation).
    The sums are what the dense all-reduce returns, up to the order in which a ring adds the ranks' contributions (bit-identical for two
    ranks, and for any exactly representable sums).  Falls back to the dense path when the union covers more than `dense_above` of the
    rows (the index gathers then cost more than the zeros).  Needs one host read of the union's size (the buffer's shape): the exchange
    sits at the step boundary.  -> {"rows_union", "rows_total", "bytes_payload", "bytes_dense_equivalent"} of this call."""
    grads = [g for g in grads if g is not None and g.numel() > 0]
    P = int(visible.numel())
    dense_bytes = sum(g.numel() * g.element_size() for g in grads)
    info = {"rows_union": P, "rows_total": P, "bytes_payload": dense_bytes, "bytes_dense_equivalent": dense_bytes}
    if not _exchange_wanted(group) or not grads:
        return info
    assert all(g.shape[0] == P for g in grads), "every gradient tensor must have one row per Gaussian"
    union = union_of_visible_rows(visible, group)
    idx = union.nonzero(as_tuple=False).reshape(-1)          # (host sync: the compacted buffer's size)
    n = int(idx.numel())
    info["rows_union"] = n
    bitmap_bytes = (P + 7) // 8
    if n > dense_above * P:
        allreduce_gradients(grads, group=group)
        info["bytes_payload"] = dense_bytes + bitmap_bytes
    elif n > 0:
        cols = [g.reshape(P, -1) for g in grads]
        buf = torch.cat([c.index_select(0, idx) for c in cols], dim=1).contiguous()
        with STALLS.waiting(buf.is_cuda):
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for g, c in zip(grads, cols):
            k = c.shape[1]
            if c.data_ptr() == g.data_ptr() and c.is_contiguous():
                c.index_copy_(0, idx, buf[:, off:off + k])
            else:   # (a non-contiguous gradient: reshape made a copy)
                c.index_copy_(0, idx, buf[:, off:off + k]); g.copy_(c.view_as(g))
            off += k
        info["bytes_payload"] = buf.numel() * buf.element_size() + bitmap_bytes
    else:
        info["bytes_payload"] = bitmap_bytes
    WIRE["bytes_payload"] += info["bytes_payload"]; WIRE["bytes_dense_equivalent"] += dense_bytes
    WIRE["rows_union"] += n; WIRE["rows_total"] += P; WIRE["calls"] += 1
    return info


class StallClock:
    """Event pairs on the CURRENT (compute) stream around the points where it waits for a collective.  With RCCL `work.wait()` only makes the
    compute stream depend on the communication stream (the host returns at once), so host wall clocks see nothing; the distance between
    an event recorded just before the wait and one recorded just after it is the time the compute stream stood still for communication --
    the EXPOSED part of the exchange (with gloo, where wait() blocks the host, the same pair spans the idle gap).  bench.py reads
    `drain_ms()` after a synchronize; two event records cost ~10 us of stream time per wait point.  Off unless `enabled` is set."""

    def __init__(self):
        self.pairs = []
        self.enabled = False   # a diagnostic: whoever switches it on (bench.py) must drain it, or the pairs pile up over a training run

    @contextlib.contextmanager
    def waiting(self, enabled: bool = True):
        if not (self.enabled and enabled and torch.cuda.is_available()):
            yield
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        try:
            yield
        finally:
            b.record()
            self.pairs.append((a, b))

    def drain_ms(self) -> float:
        """Sum of the recorded stalls (ms) and forget them; call after torch.cuda.synchronize()."""
        total = sum(a.elapsed_time(b) for a, b in self.pairs)
        self.pairs = []
        return float(total)


# process-wide clock of the compute-stream stalls caused by this module's collectives (bench.py: exposed exchange time per step)
STALLS = StallClock()


class ShExchange:
    """State of one factored SH-gradient exchange (see `factored_sh_exchange`): created by the context manager, picked up by the
    operator's FORWARD (which runs on the caller's thread, inside the context) and carried to its backward on the autograd node --
    autograd runs the backward on a thread of its own, which must not have to consult any process-wide or thread-local state."""

    def __init__(self, group=None, expand: Optional[Callable] = None, all_campos: Optional[torch.Tensor] = None,
                 reduce_all: bool = False, frames_per_rank: int = 1):
        self.group, self.expand, self.all_campos, self.reduce_all = group, expand, all_campos, reduce_all
        self.frames_per_rank = int(frames_per_rank)
        assert self.frames_per_rank >= 1
        assert not (reduce_all and self.frames_per_rank > 1), \
            "with several frames per rank the other gradients accumulate locally first: all-reduce them after the last backward (finish())"
        self.calls = 0          # exchanges completed (tests / diagnostics)
        self.bytes_sent = 0     # payload bytes this rank contributed to the collectives
        self.early_starts = 0   # all-gathers put on the wire between K7 and K8 (sr_backward_colors)
        self.exchange_ms = 0.0  # host-side wall time spent waiting on the collectives (diagnostic; bench.py reports it)
        self.attached = 0       # forward calls that attached themselves to this exchange (`attach`): frame j = the j-th forward
        self._early = {}        # frame -> (gc, gathered buffer, handle) of an all-gather started between K7 and K8
        # frame -> (gathered [world,P,3] buffer, handle, that frame's camera position).  Indexed by the frame number the FORWARD
        # drew, not by backward order: one loss over K frames runs the K backward nodes in reverse creation order.
        self._held = {}

    def attach(self) -> int:
        """Called by the operator's forward: this call is frame `attached` (mod frames_per_rank) of the step -- the row of
        `all_campos[rank]` that holds its camera.  The number rides on the autograd node to `start` / `run`."""
        j = self.attached % self.frames_per_rank
        self.attached += 1
        return j

    def _gather(self, gc: torch.Tensor):
        world = dist.get_world_size(self.group)
        flat = torch.empty(world * gc.numel(), dtype=gc.dtype, device=gc.device)   # 1-D in/out: accepted by RCCL and gloo alike
        h = dist.all_gather_into_tensor(flat, gc.reshape(-1), group=self.group, async_op=True)
        self.bytes_sent += gc.numel() * gc.element_size()
        return flat, h

    def start(self, gc: torch.Tensor, frame: int = 0) -> None:
        """Called by the operator's backward BETWEEN its two halves (sr_backward_blend / sr_backward_colors done, sr_backward_geometry
        not yet launched): `gc` [P,3] is final, so its all-gather goes on the wire now and overlaps K8.  `run` picks it up."""
        if not gc.is_contiguous():
            return
        flat, h = self._gather(gc)
        self._early[frame] = (gc, flat, h)
        self.early_starts += 1

    def _cameras(self, frames, device) -> torch.Tensor:
        """Camera position of every view in the order of the gathered gradients: view (frame j, rank r) -> row j * world + r."""
        world, K = dist.get_world_size(self.group), len(frames)
        if self.all_campos is not None:      # every rank knows the camera list: [world, 3] or [world, frames_per_rank, 3]
            cams = self.all_campos.to(device=device, dtype=torch.float32).reshape(world, K, 3)
            # the replicated list must agree with the cameras the frames were rendered with (row j = the j-th FORWARD of the step)
            # (checked the first time a given camera list is used, and with SURFEL_EXCHANGE_DEBUG=1 on every step: the comparison reads
            # device memory back -- a host sync inside the backward, on the path where K8 is meant to overlap the all-gather)
            version = getattr(self.all_campos, "_version", 0)
            seen = _CAMERA_LISTS_CHECKED.get(id(self.all_campos))   # (keyed by identity: tensors compare elementwise)
            if not (seen and seen[0]() is self.all_campos and seen[1] == version) or os.environ.get("SURFEL_EXCHANGE_DEBUG") == "1":
                rank = dist.get_rank(self.group)
                mine = torch.stack([f[2] for f in frames]).detach().to(device="cpu", dtype=torch.float32)
                if not torch.allclose(cams[rank].detach().cpu(), mine, rtol=1e-5, atol=1e-6):
                    raise RuntimeError("factored_sh_exchange: all_campos[rank] does not list this rank's cameras in the order of its forward "
                                       "calls (row j must be the camera of the j-th rasterizer call inside the block)")
                if len(_CAMERA_LISTS_CHECKED) > 256:
                    _CAMERA_LISTS_CHECKED.clear()
                _CAMERA_LISTS_CHECKED[id(self.all_campos)] = (weakref.ref(self.all_campos), version)
            return cams.transpose(0, 1).reshape(K * world, 3).contiguous()
        mine = torch.stack([f[2] for f in frames]).to(device=device, dtype=torch.float32).reshape(K * 3).contiguous()
        cams = torch.empty(world * K * 3, dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(cams, mine, group=self.group)
        return cams.view(world, K, 3).transpose(0, 1).reshape(K * world, 3).contiguous()

    def run(self, gc: torch.Tensor, means3D: torch.Tensor, campos: torch.Tensor, sh_coeffs: int, degree: int,
            also_reduce: Sequence[torch.Tensor] = (), frame: Optional[int] = None) -> Optional[torch.Tensor]:
        """gc: this rank's clamp-masked dL/drgb [P,3] of one frame.  Returns dL_dsh [P,M,3] summed over all ranks' frames -- or, while
        fewer than `frames_per_rank` frames have been handed in, None (autograd: no contribution yet): that frame's all-gather is
        already on the wire and runs under the next frame's kernels; the LAST frame's call expands all world * frames_per_rank views.

        `also_reduce`: further gradient tensors to SUM all-reduce in place (frames_per_rank == 1 only); their collective is queued
        right behind the all-gather, so it runs on the communication stream while the expansion kernel runs on the compute stream.
        `frame`: the number the forward drew from `attach` (None: the next free slot, i.e. backward order = forward order) -- the
        gathered gradient is filed under it, so that view (frame j, rank r) meets camera all_campos[r, j] whatever order autograd
        runs the K backward nodes in.  Every rank must run its backward nodes in the same order (the all-gathers pair up by issue
        order): same graph on every rank, as in DDP."""
        import time
        world = dist.get_world_size(self.group)
        gc = gc.contiguous()
        if frame is None:   # a direct call (no operator forward drew a number): frames in the order of the calls
            frame = next(j for j in range(self.frames_per_rank) if j not in self._held)
            self.attached += 1
        if frame in self._held:
            raise RuntimeError(f"factored_sh_exchange: frame {frame} handed in twice before the step's {self.frames_per_rank} frames were complete")
        early = self._early.pop(frame, None)
        if early is not None and early[0] is gc:     # started by `start` between the two halves of the backward
            _, flat, h = early
        else:
            flat, h = self._gather(gc)
        self._held[frame] = (flat, h, campos.detach().reshape(3))
        if len(self._held) < self.frames_per_rank:
            return None
        frames, self._held = [self._held[j] for j in range(self.frames_per_rank)], {}
        cams = self._cameras(frames, gc.device)
        pending = []
        groups, singles = _storage_groups([t for t in also_reduce if t is not None and t.numel() > 0])
        for t in groups + singles:
            pending.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.bytes_sent += t.numel() * t.element_size()
        t0 = time.perf_counter()
        with STALLS.waiting(gc.is_cuda):
            for _, hh, _ in frames:
                hh.wait()
        self.exchange_ms += (time.perf_counter() - t0) * 1e3
        self.calls += 1
        gathered = (frames[0][0] if len(frames) == 1 else torch.cat([f[0] for f in frames])).view((len(frames) * world,) + tuple(gc.shape))
        expand = self.expand
        if expand is None:
            from diff_surfel_rasterization import _C   # the HIP kernel; no CPU path
            expand = _C.sh_gradient_expand
        out = expand(means3D.detach(), cams, gathered, sh_coeffs, degree)
        t0 = time.perf_counter()
        with STALLS.waiting(gc.is_cuda and bool(pending)):
            for w in pending:
                w.wait()
        self.exchange_ms += (time.perf_counter() - t0) * 1e3
        return out

    def finish(self, grads: Sequence[torch.Tensor]) -> None:
        """frames_per_rank > 1: after the last frame's backward, SUM all-reduce the locally accumulated gradients of everything but the
        SH coefficients (one collective over the flat buffer the operator's backward carved them from)."""
        assert not self._held, f"{len(self._held)} of {self.frames_per_rank} frames handed in: the SH gradient of this step was never expanded"
        import time
        t0 = time.perf_counter()
        allreduce_gradients(grads, group=self.group)
        self.exchange_ms += (time.perf_counter() - t0) * 1e3
        self.bytes_sent += sum(g.numel() * g.element_size() for g in grads if g is not None)


# The exchange a forward call should attach to its autograd node.  A ContextVar: scoped to the `with` block of the calling thread /
# task, never consulted from the backward (the node carries the object itself).
_ACTIVE_SH_EXCHANGE: contextvars.ContextVar = contextvars.ContextVar("surfel_sh_exchange", default=None)


@contextlib.contextmanager
def factored_sh_exchange(group=None, expand: Optional[Callable] = None, all_campos: Optional[torch.Tensor] = None,
                         reduce_all: bool = False, frames_per_rank: int = 1):
    """Rasterizer calls made (FORWARD) within this context return, from their backward, dL_dsh ALREADY SUMMED over the
    frame-parallel ranks; the backward itself may run anywhere, later, on autograd's own thread.

    Each rank's backward all-gathers its clamp-masked colour gradient (12 B/Gaussian; plus the 12-B camera position
    unless `all_campos` [world,3] is given) and expands the SH adjoint of all `world` frames locally, instead of
    all-reducing the 192-B/Gaussian dL_dsh afterwards.  Requirements: every rank runs the same sequence of backward
    calls on the same Gaussians (same P, same subset) -- the collective sits inside backward, as in DDP.  Gradients of
    everything else (means3D, opacity, scales, rotations) stay local: pass those, and NOT the SH parameters'
    gradients, to `allreduce_gradients`.  With `reduce_all=True` the backward also all-reduces those four (one collective
    over the flat buffer they are carved from, overlapped with the expansion kernel) and EVERY parameter gradient of the
    operator leaves backward already summed -- then nothing of it may be all-reduced again.
    `frames_per_rank=K` (gradient accumulation, the lever against the serial compute -> exchange dependency): K rasterizer calls per
    rank and step attach to this exchange; the colour gradients of frame j travel while frame j + 1 is computed, the backward of the
    K-th frame returns dL_dsh of all K * world views (the earlier ones return None for it), and `ShExchange.finish(grads)` all-reduces
    the rest once.  `all_campos` is then [world, K, 3].
    No-op when torch.distributed is not initialised or the group has one rank."""
    ex = ShExchange(group, expand, all_campos, reduce_all, frames_per_rank) if _exchange_wanted(group) else None
    token = _ACTIVE_SH_EXCHANGE.set(ex)
    try:
        yield ex
    finally:
        _ACTIVE_SH_EXCHANGE.reset(token)
        if ex is not None and ex.attached == 0:
            # the binding happens at FORWARD time: a block around loss.backward() alone exchanges nothing and the ranks diverge silently
            import warnings
            warnings.warn("factored_sh_exchange: no rasterizer forward ran inside the block -- wrap the render call (the forward), not only "
                          "loss.backward(); this step's SH gradients were NOT exchanged", RuntimeWarning, stacklevel=3)


def active_sh_exchange() -> Optional[ShExchange]:
    """The exchange of the enclosing `factored_sh_exchange` block on THIS thread (the operator's forward stores it on its autograd node)."""
    return _ACTIVE_SH_EXCHANGE.get()


def reduce_densification_stats(viewspace_grad: torch.Tensor, radii: torch.Tensor, xyz_gradient_accum: torch.Tensor,
                               denom: torch.Tensor, max_radii2D: torch.Tensor, group=None) -> None:
    """Per-rank contribution of one rendered view, then cross-rank SUM / SUM / MAX.

    Mirrors GaussianModel.add_densification_stats + the max_radii2D update for this rank's frame and makes
    the three statistics identical on every rank [REF /root/reference/scene/gaussian_model.py:555-557; train.py:168-169].
    ONE packed buffer per rank -- (|viewspace grad|, radius) = 8 B per Gaussian, 0 / 0 where the view does not see it; `denom`'s
    contribution is radius > 0 and needs no slot -- and no boolean-index kernels.  Up to three ranks it travels as ONE all-gather
    (every rank then sums / counts / maximises the `world` rows locally: (N - 1) * 8 B received per Gaussian, against 12 B for
    reducing three arrays); from four ranks on as ONE all-reduce pair issued together, SUM over the (norm, visible) columns and MAX
    over the radii (2 (N - 1) / N * 12 B, which the all-gather's (N - 1) * 8 B exceeds)."""
    vis = radii > 0
    norm = torch.where(vis, torch.norm(viewspace_grad, dim=-1), torch.zeros((), dtype=viewspace_grad.dtype, device=viewspace_grad.device))
    rad = torch.where(vis, radii, torch.zeros_like(radii)).to(torch.float32)   # radii are small integers: exact in float32
    if not _exchange_wanted(group):
        acc, cnt, mx = norm, vis.to(norm.dtype), rad
    elif dist.get_world_size(group) <= 3:
        world = dist.get_world_size(group)
        mine = torch.stack([norm.to(torch.float32), rad]).contiguous()            # [2, P]
        rows = torch.empty(world * mine.numel(), dtype=torch.float32, device=mine.device)
        dist.all_gather_into_tensor(rows, mine.reshape(-1), group=group)
        rows = rows.view(world, 2, -1)
        acc, cnt, mx = rows[:, 0].sum(0), (rows[:, 1] > 0).sum(0).to(torch.float32), rows[:, 1].max(0).values
    else:
        sums = torch.stack([norm.to(torch.float32), vis.to(torch.float32)]).contiguous()   # [2, P]: ONE SUM collective for both
        h1 = dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group, async_op=True)
        h2 = dist.all_reduce(rad, op=dist.ReduceOp.MAX, group=group, async_op=True)
        h1.wait(); h2.wait()
        acc, cnt, mx = sums[0], sums[1], rad
    xyz_gradient_accum += acc.to(xyz_gradient_accum.dtype).view_as(xyz_gradient_accum)
    denom += cnt.to(denom.dtype).view_as(denom)
    torch.maximum(max_radii2D, mx.to(max_radii2D.dtype).view_as(max_radii2D), out=max_radii2D)
