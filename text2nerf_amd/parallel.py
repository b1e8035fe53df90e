"""Ray-tile sharding across the GPUs of one node (one process per GPU, torch.distributed: backend "nccl" = RCCL on
ROCm, "gloo" in the CPU tests). Rays are independent (the reference itself splits them into arbitrary chunks,
renderer.py:32-33), so there is no data-path collective inside the render; the only exchange is the all-gather of the
rendered [rays, 4] (rgb + depth) tiles. The field (69.6 MB) is replicated per GPU.

On a fully connected 8-GPU xGMI node each rank's tile (C4: 320 000 rays x 16 B = 5.1 MB) goes to its 7 peers over 7
separate links; one ``all_gather_into_tensor`` of equal-sized tiles lets RCCL pick that direct algorithm, so tiles are
padded to equal length instead of issuing per-rank variable-size sends."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n_rays: int, world: int, rank: int):
    """Contiguous, balanced ray tile of `rank`: the first n_rays % world ranks carry one extra ray."""
    base, rem = divmod(n_rays, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def tile_capacity(n_rays: int, world: int) -> int:
    return (n_rays + world - 1) // world


def all_gather_tiles(tile: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather equal-sized [n, C] tiles -> [world * n, C] on every rank (one collective)."""
    world = dist.get_world_size(group)
    out = torch.empty((world * tile.shape[0],) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
    dist.all_gather_into_tensor(out, tile.contiguous(), group=group)
    return out


def band_layout(n_rays: int, frame_width: int, world: int, band_rows: int = 8):
    """Interleaved assignment of a row-major frame to `world` ranks in bands of `band_rows` image rows (border and
    centre rays cross different lengths of the box, so contiguous tiles are unevenly loaded; a rank that takes every world-th band
    sees the same mix as its peers). Returns ``(bands_per_rank, band_rays)`` when the frame divides evenly — whole rows, a multiple
    of ``world * band_rows`` of them — else ``None`` (the caller then uses contiguous tiles). ``band_rows = 8`` keeps every band a
    whole row of the tile marcher's 8x8-pixel tiles."""
    if not frame_width or frame_width <= 0 or n_rays % frame_width:
        return None
    rows = n_rays // frame_width
    if rows % (world * band_rows):
        return None
    return rows // (world * band_rows), band_rows * frame_width


def band_shard(rays: torch.Tensor, frame_width: int, world: int, rank: int, band_rows: int = 8):
    """The rays of `rank` under band_layout: bands rank, rank + world, ... concatenated in image order ([R / world, C])."""
    per, n = band_layout(rays.shape[0], frame_width, world, band_rows)
    return rays.view(per, world, n, rays.shape[1])[:, rank].reshape(per * n, rays.shape[1])


def band_unshard(gathered: torch.Tensor, frame_width: int, world: int, band_rows: int = 8):
    """Inverse of band_shard on the all-gathered ``[world * R / world, C]`` tensor (rank-major) -> image order ``[R, C]``."""
    R = gathered.shape[0]
    per, n = band_layout(R, frame_width, world, band_rows)
    return gathered.view(world, per, n, gathered.shape[1]).permute(1, 0, 2, 3).reshape(R, gathered.shape[1])


def render_sharded(rays: torch.Tensor, render_fn, group=None, frame_width: int = 0):
    """Render `rays` [R, >=6] cooperatively: each rank renders its tile with `render_fn(rays_tile) -> (rgb [n,3], depth [n])`, then
    the tiles are all-gathered (ONE collective of equal-sized [n, 4] tiles). Returns (rgb [R,3], depth [R]) on every rank, bitwise
    equal to the single-GPU result because every ray is computed by exactly one rank with the same kernels.
    `frame_width` > 0 states that `rays` is a row-major frame of that width: ranks then take interleaved 8-row bands
    (band_layout) when the frame divides evenly, contiguous tiles otherwise."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    R = rays.shape[0]
    if band_layout(R, frame_width, world) is not None:
        rgb, depth = render_fn(band_shard(rays, frame_width, world, rank).contiguous())
        tile = torch.cat([rgb, depth[:, None]], 1)
        out = band_unshard(all_gather_tiles(tile, group), frame_width, world)
        return out[:, :3].contiguous(), out[:, 3].contiguous()
    lo, hi = shard_bounds(R, world, rank)
    cap = tile_capacity(R, world)
    rgb, depth = render_fn(rays[lo:hi])
    tile = torch.zeros(cap, 4, dtype=rgb.dtype, device=rgb.device)
    tile[: hi - lo, :3] = rgb
    tile[: hi - lo, 3] = depth
    full = all_gather_tiles(tile, group).view(world, cap, 4)
    parts = []
    for r in range(world):
        l, h = shard_bounds(R, world, r)
        parts.append(full[r, : h - l])
    out = torch.cat(parts, 0)
    return out[:, :3].contiguous(), out[:, 3].contiguous()


# ---- data-parallel training (SURVEY.md §8 e, C5) -----------------------------------------------------------------------------
# The 16 384-ray batch of text2nerf_main.py:547-601 is split into equal contiguous shards (one per GPU); every rank runs the
# HIP forward/backward on its shard, then ONE all-reduce sums the whole 17.4 M-element fp32 gradient (69.6 MB) as a single
# flat message — on point-to-point xGMI the per-link bandwidth, not the launch count, is the cost, so one large
# collective beats per-tensor ones (19 tensors, most of them tiny). The TV regulariser does not depend on the rays: its
# gradient is identical on every rank and is added AFTER the all-reduce (or by the fused TVAdam step), not reduced.

def shard_batch(n: int, world: int, rank: int):
    """Equal-sized contiguous shard [lo, hi) of an n-ray batch (n is truncated to a multiple of world, like DistributedSampler
    with drop_last): equal shards keep mean-reduced losses consistent with the single-GPU batch mean."""
    per = n // world
    return rank * per, (rank + 1) * per


_BUCKET_STREAMS = {}


def allreduce_buckets(buf: torch.Tensor, split: int, group=None, first_ready=None):
    """All-reduce (sum) the flat buffer `buf` in place as TWO messages, [0, split) and [split, n): the same element-wise sums as one
    flat message (bitwise: every element is reduced with the same operands either way on a two-rank group; on larger groups the ring's
    chunking may differ). `first_ready(stream)`: called with a side stream on which the first bucket's collective is then issued — the
    caller makes that stream wait for whatever finishes the first bucket EARLY (the density half of the factor gradients is final
    ~0.5 ms before the appearance half: its 17 MB are on the links while the appearance scatter still runs). Without it both go out on
    the current stream, back to back."""
    n = buf.numel()
    split = max(0, min(int(split), n))
    if split == 0 or split == n:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        return
    a, b = buf[:split], buf[split:]
    if first_ready is not None and buf.is_cuda:
        key = str(buf.device)
        side = _BUCKET_STREAMS.get(key)
        if side is None:
            side = _BUCKET_STREAMS[key] = torch.cuda.Stream(buf.device)
        first_ready(side)
        with torch.cuda.stream(side):
            wa = dist.all_reduce(a, op=dist.ReduceOp.SUM, group=group, async_op=True)
        wb = dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group, async_op=True)
        wa.wait()      # (stream-level for RCCL: the CURRENT stream waits for both collectives; gloo blocks the host)
        wb.wait()
        torch.cuda.current_stream(buf.device).wait_stream(side)
    else:
        wa = dist.all_reduce(a, op=dist.ReduceOp.SUM, group=group, async_op=True)
        wb = dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group, async_op=True)
        wa.wait()
        wb.wait()


def allreduce_gradients(params, group=None, average=True, field=None, overlap=True):
    """Sum (or average) the gradients over the ranks. `field` with ``defer_factor_grads`` set (optim.TVAdam(field=...)): the 12 plane /
    line gradients are all-reduced IN PLACE in the field's contiguous channel-last buffer (69.6 MB at 300^3, no concatenation or
    copy-back) as two buckets — density (17.5 MB, issued on a side stream as soon as the backward's density scatter is done:
    t2n_field_wait_density_grads) and appearance (52 MB, behind the whole backward); `overlap=False`: one flat message behind the
    backward. The remaining (head) tensors travel as one small flat message. Without `field`: one flat all-reduce of the .grad of
    `params`; a None grad counts as zero on this rank (every rank must pass the same parameter list)."""
    world = dist.get_world_size(group)
    deferred = field is not None and bool(getattr(field, "defer_factor_grads", False)) and field.supports_deferred_factor_grads()
    if deferred:
        buf = field.factor_grad_buffer()
        if overlap:
            from . import _lib
            lib = _lib.load()
            h = field.sync_params()
            nden = int(lib.t2n_field_grad_buffer_density_bytes(h)) // 4
            allreduce_buckets(buf, nden, group, first_ready=lambda st: _lib.check(
                lib.t2n_field_wait_density_grads(h, st.cuda_stream), "t2n_field_wait_density_grads"))
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        if average:
            buf.div_(world)
        field._gbuf_dirty = True
        field._gbuf_reduced = True      # TVAdam(field=...) refuses to step from un-reduced gradients when world > 1
        fac = {id(p) for p in field._all_params()[:12]}
        params = [p for p in params if id(p) not in fac]
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    fs = field.__dict__.get("_fused_step") if field is not None else None
    if fs is not None and fs.owns_head_grads(params):
        # the fused step (text2nerf_amd/trainer.py) keeps the head gradients in ONE flat buffer with a vote word behind them (non-zero:
        # some rank's appearance rows did not fit its capacity — the optimiser phase then withholds the update on EVERY rank): reduced in
        # place, no concatenation, no copy back
        dist.all_reduce(fs.head_grads, op=dist.ReduceOp.SUM, group=group)
        if average:
            fs.head_grads.div_(world)
        return
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat.div_(world)
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n


def broadcast_parameters(params, src=0, group=None):
    """Make every rank start from rank `src`'s parameters (one flat broadcast)."""
    params = list(params)
    if not params:
        return
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for p in params:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))   # in-place on the parameter itself: bumps its version, so the native field
            off += n                                 # re-uploads (a write through .data would leave a stale device copy)


# ---- sharded optimiser (SURVEY.md §8 e) -----------------------------------------------------------------------
# With the flat all-reduce above every rank repeats TV + Adam over all 70 MB of factors (10 streams of HBM traffic per element) and
# receives gradients it only needs 1 / world of. Here each plane's 64-position blocks are split evenly over the ranks (the "body"; the few
# blocks behind it, the six lines and the head stay replicated):
#   phase 1 (t2n_train_step, shard_world / shard_rank)   the owner seeds world x TV, everybody else zero, then the local backward
#   reduce()    reduce-scatter(AVG) of each body IN PLACE (a rank receives only its slice: (world - 1) / world of the bytes of an
#               all-reduce's reduce-scatter half, and no all-gather half), one small all-reduce(AVG) of replicated blocks + lines + head
#   phase 2     TV-seeded Adam on the owned + replicated blocks: 1 / world of the optimiser traffic; moments of other blocks never touched
#   gather()    all-gather of each body of the channel-last parameter copies IN PLACE
#   phase 4     the gathered blocks -> the caller's reference-layout tensors
# Bytes on the links per rank and step at world = 8, 300^3: reduce-scatter 7/8 x 69 MB + all-gather 7/8 x 69 MB — the same as the flat
# all-reduce's — but TV + Adam drop from ~150 us to ~20 us per rank, and the all-gather of the parameters overlaps nothing it has to wait
# for (the next step's march needs the density planes only: issued first).

def shard_layout(grid, world, n_den=16, n_app=48):
    """Python mirror of t2n_field_shard_layout (text2nerf_amd/csrc/t2n_optim.hip shard_partition): rows ``[offset, slice, total]`` in
    floats for the 12 factor tensors (density planes, density lines, appearance planes, appearance lines) of the channel-last gradient
    buffer; `slice` = floats of ONE rank's part of the body (0: wholly replicated)."""
    MAT = ((0, 1), (0, 2), (1, 2))
    VEC = (2, 1, 0)
    out, off = [], 0
    for q in range(4):
        C = n_den if q < 2 else n_app
        for k in range(3):
            n = int(grid[MAT[k][1]]) * int(grid[MAT[k][0]]) if q % 2 == 0 else int(grid[VEC[k]])
            chunk = ((n // 64) // world) if (q % 2 == 0 and world > 1) else 0
            out.append([off // 4, chunk * 64 * C, n * C])
            off += (n * C * 4 + 255) // 256 * 256
    return out


class _DeviceView:
    """A device allocation of the native library as an object torch can wrap without a copy (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class ShardedExchange:
    """The two exchanges of the sharded optimiser, as the ``all_reduce`` argument of ``train_step``::

        ex = ShardedExchange.for_field(field, group)
        field.train_step(rays, rgb, depth, optimizer, ..., all_reduce=ex)

    `layout`: shard_layout rows; `grads`: the flat channel-last gradient buffer; `params`: the 12 channel-last parameter arrays (flat);
    `extra`: callable returning further flat tensors averaged with the replicated parts (the fused step's head gradients + vote word)."""

    def __init__(self, layout, grads, params, group=None, extra=None, world=None, rank=None):
        self.group = group
        if world is None:       # (world / rank given: a subclass brings its own collectives — tests/helpers/virtual_ranks.py)
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            self._nccl = dist.get_backend(group) == "nccl"
        self.world, self.rank = int(world), int(rank)
        self.layout, self.grads, self.params, self.extra = layout, grads, params, extra

    @classmethod
    def for_field(cls, field, group=None, world=None, rank=None):
        from . import _lib
        import ctypes as C
        lib = _lib.load()
        if world is None:
            world = dist.get_world_size(group)
        grads = field.factor_grad_buffer(_raw=True)
        h = field.sync_params()
        lay = (C.c_int64 * 36)()
        _lib.check(lib.t2n_field_shard_layout(h, world, lay), "t2n_field_shard_layout")
        layout = [[int(lay[3 * t]), int(lay[3 * t + 1]), int(lay[3 * t + 2])] for t in range(12)]
        params = []
        for t in range(12):
            p = C.c_void_p()
            _lib.check(lib.t2n_field_factor_buffer(h, t, C.byref(p)), "t2n_field_factor_buffer")
            params.append(torch.as_tensor(_DeviceView(p.value, layout[t][2]), device=grads.device))
        ex = cls(layout, grads, params, group, world=None if rank is None else world, rank=rank)
        ex.field = field
        return ex

    def __call__(self):
        raise RuntimeError("ShardedExchange drives the fused train step (TensorVMSplit.train_step with optim.TVAdam(field=...)); "
                           "use parallel.allreduce_gradients for the composed step")

    # -- collectives with an averaging reduction; gloo (CPU tests) has neither reduce_scatter nor AVG: all-reduce + divide there
    def _reduce_scatter_avg(self, own, body):
        if self._nccl:
            dist.reduce_scatter_tensor(own, body, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(body, op=dist.ReduceOp.SUM, group=self.group)
            own.div_(self.world)

    def _all_reduce_avg(self, t):
        if self._nccl:
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t.div_(self.world)

    def _all_gather(self, body, own):
        if self._nccl:
            dist.all_gather_into_tensor(body, own, group=self.group)
        else:
            parts = [torch.empty_like(own) for _ in range(self.world)]
            dist.all_gather(parts, own.clone(), group=self.group)
            for r, part in enumerate(parts):
                body[r * own.numel():(r + 1) * own.numel()].copy_(part)

    def _slices(self, flat, t, base):
        off = self.layout[t][0] if base else 0
        sl, total = self.layout[t][1], self.layout[t][2]
        body = flat[off:off + self.world * sl]
        return body, body[self.rank * sl:(self.rank + 1) * sl], flat[off + self.world * sl:off + total]

    def reduce(self, fused_step=None):
        """Average the ranks' gradient buffers: every body by reduce-scatter (this rank's slice ends up averaged, the other slices
        are garbage nobody reads), the replicated remainders + `extra` by ONE small all-reduce."""
        g = self.grads
        small = []
        for t in range(12):
            body, own, rest = self._slices(g, t, True)
            if own.numel():
                self._reduce_scatter_avg(own, body)
            if rest.numel():
                small.append(rest)
        extra = []
        if fused_step is not None:
            extra = [fused_step.head_grads]
        elif self.extra is not None:
            extra = list(self.extra())
        pieces = small + extra
        flat = torch.cat([x.reshape(-1) for x in pieces])
        self._all_reduce_avg(flat)
        off = 0
        for x in pieces:
            n = x.numel()
            x.copy_(flat[off:off + n].view_as(x))
            off += n
        f = getattr(self, "field", None)
        if f is not None:
            f._gbuf_dirty, f._gbuf_reduced = True, True

    def gather(self, fused_step=None):
        """All-gather every body of the channel-last parameter copies in place (density planes first: the next march reads them)."""
        for t in range(12):
            body, own, _ = self._slices(self.params[t], t, False)
            if own.numel():
                self._all_gather(body, own)
