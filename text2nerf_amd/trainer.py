"""The fused training step: one optimisation step of text2nerf_main.py:547-601 as ONE submission (t2n_train_step / its hipGraph).

``FusedStep`` is what ``TensorVMSplit.train_step`` runs on when the field has the tuned shape (fused MLP_Fea_noview head, 16 + 48
components, fp32 factor storage) and the optimiser is ``optim.TVAdam(field=tensorf)``:

* the batch (rays, the CPU-generator jitter draws of models/tensorBase.py:313-317, colours, depths) and the step's scalars (the 19
  learning rates, the two TV weights) travel as ONE staged host -> device copy into one of four fixed input buffers;
* the C call behind it (``t2n_train_step``) enqueues TV gradient, train-mode render, the driver's loss, the backward, Adam on all 19
  tensors and the re-pack of the head operands on four streams, reads nothing back and takes every per-step quantity from device memory
  — so ``graph=True`` captures it ONCE per input buffer into a hipGraph (``t2n_train_graph_capture``) and a step is one
  ``hipGraphLaunch``;
* the appearance-row CAPACITY of a step is a guess (1.25 x the largest need of the last steps). A step that needs more applies NO
  update: every optimiser kernel reads the verdict from device memory and returns, Adam's step count does not advance. The host learns
  it from a record in pinned memory, without waiting, and submits the same batch again with a capacity that holds it — no truncated
  gradient is ever applied. ``device_rows_overflows`` counts those steps.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import T2NError

_RING = 4          # input buffers (and graphs); the host runs at most two steps ahead of the device, so a slot's verdict is known before reuse
_RUN_AHEAD = 2
PIPELINE_MAX_BYTES = 3 << 30


def _ladder(n: int, step: float = 1.25, floor: int = 1024) -> int:
    v = floor
    while v < n:
        v = int(v * step) + 31
    return v // 32 * 32


class FusedStep:
    def __init__(self, field, optimizer):
        if getattr(optimizer, "field", None) is not field:
            raise T2NError("fused train step: needs optim.TVAdam(field=tensorf)")
        self.field, self.opt = field, optimizer
        self.dev = field.basis_mat.weight.device
        self.issued = 0            # t2n_train_step submissions (= the device's sequence counter)
        self.seen = 0              # records consumed
        self.rows_cap = 0
        self.needs = []            # recent row needs
        self.margin = 1.25
        self.slots = [None] * _RING    # per input buffer: dict(seq, cap, args, meta) of the step it holds
        self.inbuf = [None] * _RING
        self.pinned = [None] * _RING
        self.graphs = {}           # (slot, key) -> graph handle
        self.ws = None
        self.events = []
        self.head_grads = torch.zeros(_lib.TRAIN_HEAD_GRAD_FLOATS, device=self.dev)
        self.losses = torch.zeros(4, device=self.dev)
        self.replays = 0
        self.copy_stream = None
        self.cap_once = 0
        self.host_batch = None          # (pinned pointer, bytes) of the staged batch when the C call is to copy it
        self.pipe_ok = False
        self.pipe_ws = False
        self.pipelined_launches = 0
        self.pipeline = True            # (False: always the serial form)
        self._key_after = None
        self.slot_done = [None] * _RING
        self.queue = []            # withheld batches waiting for their replay: (input buffer copy, meta, rows needed)
        self.graph_launches = 0
        self.graph_captures = 0
        self.eager_launches = 0
        self._handle_seen = None
        self._mom_cache = None
        self._arg_cache = {}
        self._group_of = None
        self._pin_np = [None] * _RING
        self._ws_for = None

    # ---- optimiser state ------------------------------------------------------------------------------------------------------------
    def _moments(self):
        f, opt = self.field, self.opt
        c = self._mom_cache
        if c is not None:
            # the host side of a 2 048-ray step is as long as its device side: the moment tensors are looked up once; a replaced optimiser
            # state (load_state_dict) shows in the first / last entries
            ps, ms, vs = c
            st0, st18 = opt.state[ps[0]], opt.state[ps[18]]
            if st0.get("exp_avg_cl") is ms[0] and st18.get("exp_avg") is ms[18] and st0["step"] == st18["step"]:
                return ps, ms, vs, int(st0["step"])
        ps = f._all_params()
        ms, vs, steps = [], [], []
        fresh = [i for i, p in enumerate(ps[:12]) if "exp_avg_cl" not in opt.state[p]]
        pool = None
        if len(fresh) == 12:
            # a new optimiser: the 24 channel-last moment buffers of the factor tensors are slices of ONE allocation the field keeps across
            # optimisers (zeroed here). Fresh allocations per optimiser land wherever the caching allocator has room, and the step time
            # follows the placement: 0.86 against 0.99 ms per 16 384-ray step, alternating from one optimiser to the next (measured)
            n = sum(p.numel() for p in ps[:12])
            pool = f.__dict__.get("_fused_moments")
            if pool is None or pool.numel() != 2 * n or pool.device != ps[0].device:
                pool = f.__dict__["_fused_moments"] = torch.empty(2 * n, device=ps[0].device, dtype=torch.float32)
            pool.zero_()
        off = 0
        for i, p in enumerate(ps):
            st = opt.state[p]
            a, b = ("exp_avg_cl", "exp_avg_sq_cl") if i < 12 else ("exp_avg", "exp_avg_sq")
            if a not in st:
                st.setdefault("step", 0)
                if pool is not None and i < 12:
                    st[a] = pool[off:off + p.numel()]
                    st[b] = pool[pool.numel() // 2 + off:pool.numel() // 2 + off + p.numel()]
                else:
                    st[a] = torch.zeros(p.numel(), device=p.device, dtype=torch.float32) if i < 12 else torch.zeros_like(p)
                    st[b] = torch.zeros(p.numel(), device=p.device, dtype=torch.float32) if i < 12 else torch.zeros_like(p)
            if i < 12:
                off += p.numel()
            ms.append(st[a]); vs.append(st[b]); steps.append(int(st["step"]))
        if len(set(steps)) != 1:
            raise T2NError("fused train step: the 19 tensors must share one Adam step count")
        self._mom_cache = (ps, ms, vs)
        self._arg_cache = {}
        return ps, ms, vs, steps[0]

    def _hyper(self, tv):
        """The step's scalars as a list of TRAIN_HYPER_FLOATS floats (19 learning rates, the two TV weights x 1e-2), betas, eps."""
        f, opt = self.field, self.opt
        groups = opt.param_groups
        gi = self._group_of
        if gi is None or len(gi[1]) != len(groups):
            where = {id(p): k for k, g in enumerate(groups) for p in g["params"]}
            ps = f._all_params()
            if any(id(p) not in where for p in ps):
                raise T2NError("fused train step: every tensor of the field must be in a parameter group")
            gi = self._group_of = ([where[id(p)] for p in ps], list(groups))
        lrs = [float(g["lr"]) for g in groups]
        h = [lrs[k] for k in gi[0]]
        tv_d, tv_a = f._tv_weights(tv)
        h += [tv_d, tv_a] + [0.0] * (_lib.TRAIN_HYPER_FLOATS - 21)
        g0 = groups[0]
        b0, e0 = g0["betas"], g0["eps"]
        for g in groups:
            if g["betas"] != b0 or g["eps"] != e0:
                if tuple(g["betas"]) != tuple(b0) or float(g["eps"]) != float(e0):
                    raise T2NError("fused train step: all parameter groups must share betas and eps")
        return h, (float(b0[0]), float(b0[1])), float(e0)

    # ---- the device's record ----------------------------------------------------------------------------------------------------------
    def _poll(self):
        """Consume the records that have landed: row needs -> next capacity; withheld steps -> replays. Never waits."""
        f = self.field
        if f._handle is None or not self.issued:
            return
        rec = (C.c_uint32 * 36)()
        _lib.check(_lib.load().t2n_field_train_record(f._handle, rec), "t2n_field_train_record")
        while self.seen < self.issued:
            seq = self.seen
            need, tag = int(rec[4 + 2 * (seq & 15)]), int(rec[5 + 2 * (seq & 15)])
            if (tag >> 1) != ((seq + 1) & 0x7fffffff):
                if (tag >> 1) > seq + 1:
                    raise T2NError("fused train step: the device ran more than 16 steps ahead of the host's polls")
                break                       # this step's record has not arrived yet
            withheld = tag & 1
            self.needs = (self.needs + [need])[-8:]
            slot = self.slots[seq % _RING]
            if withheld:
                f.device_rows_overflows = getattr(f, "device_rows_overflows", 0) + 1
                self.margin = min(2.0, self.margin * 1.25)
                if slot is None or slot["seq"] != seq:
                    raise T2NError("fused train step: a withheld step's batch is no longer held (internal)")
                # (a copy on the current stream: ordered in front of whatever overwrites the input buffer later)
                self.queue.append((self.inbuf[slot["i"]].clone(), slot["meta"], need))
                self.slots[seq % _RING] = None
            self.seen += 1
        if self.needs:
            want = _ladder(int(max(self.needs) * self.margin) + 64)
            if want > self.rows_cap or want < int(self.rows_cap * 0.8):     # (grids and zero fills scale with the capacity: follow it down too)
                self.rows_cap = want

    def sync(self):
        """Drain the stream, consume every record, replay what was withheld: parameters and Adam's step count are then exactly those of
        `issued` applied steps."""
        while True:
            torch.cuda.current_stream(self.dev).synchronize()
            self._poll()
            if not self.queue:
                break
            self._drain_queue()

    # ---- submission -------------------------------------------------------------------------------------------------------------------
    def _workspace(self, R, N, cap):
        if self.ws is not None and self._ws_for == (R, N, cap, self.field._handle.value):
            return self.ws
        lib = _lib.load()
        self._ws_for = (R, N, cap, self.field._handle.value)
        need = int(lib.t2n_train_step_workspace_bytes(self.field._handle, R, N, cap))
        if need == 0:
            raise T2NError("t2n_train_step_workspace_bytes: bad shape")
        # pipelined steps alternate between two halves of the workspace (include/t2n.h); a step that needs more than PIPELINE_MAX_BYTES
        # (the fog phase of a run: millions of appearance rows) runs in the serial form on one
        self.pipe_ws = need <= PIPELINE_MAX_BYTES
        if self.pipe_ws:
            need *= 2
        if self.ws is None:
            self.ws = self.field.__dict__.get("_fused_ws")     # (the field keeps ONE such buffer: a new optimiser / driver takes it over)
            if self.ws is not None and self.ws.device != self.dev:
                self.ws = None
        # grow when it does not fit; shrink when a third would do (after the fog phase of a run the rows fall back by a factor of ~6)
        if self.ws is None or self.ws.numel() < need or self.ws.numel() > 3 * need + (64 << 20):     # (hysteresis: 3 x)
            had = self.ws is not None
            self.ws = self.field.__dict__["_fused_ws"] = None
            self._drop_graphs()
            if had:
                torch.cuda.empty_cache()
            # (+ 2 MiB nobody is told about: a buffer that ends where its mapping ends turns any over-read by a few bytes into a GPU fault)
            self.ws = self.field.__dict__["_fused_ws"] = torch.empty(need + (2 << 20), dtype=torch.uint8, device=self.dev)
        return self.ws

    def _drop_graphs(self):
        lib = _lib.load()
        for g in self.graphs.values():
            lib.t2n_train_graph_destroy(g)
        self.graphs = {}

    def __del__(self):
        try:
            self._drop_graphs()
        except Exception:
            pass

    def _args(self, slot_i, R, stride, N, flags, phases, cap, betas, eps, w_depth, w_trans, delta):
        f = self.field
        ps, ms, vs, _ = self._moments()
        buf = self.inbuf[slot_i]
        ws = self._workspace(R, N, cap)
        ck = (slot_i, R, stride, N, flags, phases, cap, betas, eps, w_depth, w_trans, delta, ws.data_ptr(), buf.data_ptr(), ps[0].data_ptr())
        hit = self._arg_cache.get(ck)
        if hit is not None:
            hit[0].host_batch, hit[0].host_batch_bytes = None, 0
            return hit
        if len(self._arg_cache) > 64:
            self._arg_cache.clear()
        a = _lib.TrainStepArgs()
        o = 0
        base = buf.data_ptr()
        a.rays = base; o += R * stride
        a.jitter = base + 4 * o; o += R
        a.rgb_target = base + 4 * o; o += 3 * R
        a.depth_target = base + 4 * o; o += R
        a.hyper = base + 4 * o
        a.n_rays, a.ray_stride, a.n_samples, a.flags, a.phases = R, stride, N, flags, phases
        a.w_depth, a.w_trans, a.delta = w_depth, w_trans, delta
        a.beta1, a.beta2, a.eps = betas[0], betas[1], eps
        a.params = f._param_struct([p.detach() for p in ps])
        for i in range(19):
            a.exp_avg[i] = ms[i].data_ptr()
            a.exp_avg_sq[i] = vs[i].data_ptr()
        a.head_grads = self.head_grads.data_ptr()
        a.rows_capacity = cap
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        a.losses = self.losses.data_ptr()
        a.host_batch, a.host_batch_bytes, a.batch_buffer = None, 0, buf.data_ptr()
        key = (R, stride, N, flags, phases, cap, betas, eps, w_depth, w_trans, delta, ws.data_ptr(), buf.data_ptr(),
               tuple(p.data_ptr() for p in ps), tuple(t.data_ptr() for t in ms), tuple(t.data_ptr() for t in vs))
        self._arg_cache[ck] = (a, key)
        return a, key

    def _launch(self, slot_i, a, key, graph, m=None):
        lib = _lib.load()
        f = self.field
        st = _lib.current_stream_ptr(self.dev)
        with torch.cuda.device(self.dev):
            if graph and self.eager_launches:
                g = self.graphs.get((slot_i, key))
                if g is None:
                    # a key that changed (capacity, workspace, shapes, optimiser state) leaves the old graphs behind: dropped, and the
                    # step is captured for EVERY input buffer of the ring now (a capture costs a millisecond: not inside somebody's loop,
                    # one at a time, four steps in a row)
                    self._drop_graphs()
                    for j in range(_RING):
                        self._ensure_inbuf(j, self.inbuf[slot_i].numel())
                        aj, kj = self._args(j, m["R"], m["stride"], m["N"], m["flags"], 3, int(a.rows_capacity), m["betas"], m["eps"],
                                            m["w_depth"], m["w_trans"], m["delta"])
                        h = C.c_void_p()
                        _lib.check(lib.t2n_train_graph_capture(f._handle, C.byref(aj), st, C.byref(h)), "t2n_train_graph_capture")
                        self.graphs[(j, kj)] = h
                        self.graph_captures += 1
                    g = self.graphs[(slot_i, key)]
                    self.graph_nodes = int(lib.t2n_train_graph_nodes(g))
                _lib.check(lib.t2n_train_graph_launch(g, st), "t2n_train_graph_launch")
                self.graph_launches += 1
            else:
                if self.host_batch is not None:
                    a.host_batch, a.host_batch_bytes = self.host_batch     # (the C call copies the batch; pipelined form when it may)
                    a.flags = int(a.flags) & ~_lib.FLAG_PIPELINE
                    if self.pipe_ok and self.pipe_ws and int(a.phases) == 3:
                        a.flags = int(a.flags) | _lib.FLAG_PIPELINE
                        self.pipelined_launches += 1
                _lib.check(lib.t2n_train_step(f._handle, C.byref(a), st), "t2n_train_step")
                self.eager_launches += 1

    def _after_update(self):
        """The C call changed all 19 tensors in place (and left the device copies / packed operands current)."""
        f = self.field
        ps = self._mom_cache[0] if self._mom_cache is not None else f._all_params()
        try:
            torch._C._increment_version(ps)
        except Exception:
            from .optim import _bump_version
            for p in ps:
                _bump_version(p)
        key = tuple([(p.data_ptr(), p._version) for p in ps])
        f._uploaded_key = key
        f._device_factor_key = key[:12]
        f._gbuf_dirty = True
        f._gbuf_stale = True          # consumed: any other reader sees it zeroed first
        f._gbuf_reduced = False
        f._deferred_grad_key = None
        for p in ps:
            self.opt.state[p]["step"] = int(self.opt.state[p]["step"]) + 1

    def _drain_queue(self):
        """Submit the withheld batches again (eagerly), each with room for the rows its record said it needs."""
        while self.queue:
            data, m, need = self.queue.pop(0)
            self.rows_cap = max(self.rows_cap, _ladder(need + 32))
            i = self._free_slot()
            self._ensure_inbuf(i, data.numel())
            self.inbuf[i].copy_(data)
            for p in self.field._all_params():     # the withheld submission advanced the step-count mirrors: this one replaces it
                self.opt.state[p]["step"] = int(self.opt.state[p]["step"]) - 1
            self._submit(i, m, graph=False)
            self.replays += 1

    def _free_slot(self):
        """Input buffer of the next submission; its previous batch must have its verdict (else the host is a full ring ahead: wait)."""
        i = self.issued % _RING
        held = self.slots[i]
        if held is not None and held["seq"] >= self.seen:
            torch.cuda.current_stream(self.dev).synchronize()
            self._poll()
        return i

    def _ensure_inbuf(self, i, n):
        if self.inbuf[i] is None or self.inbuf[i].numel() != n:
            # the ring of input buffers belongs to the FIELD (like the workspace and the moment pool): a new optimiser's driver takes it over —
            # its first step drains the stream first (step(): a new driver of a handle) — instead of pinning four new host buffers, whose
            # first engine copy each stalls the stream for milliseconds (a bench block with a fresh optimiser read 0.95 instead of 0.82 ms
            # per step whenever its buffers were new)
            ring = self.field.__dict__.get("_fused_ring")
            if ring is None or ring["n"] != n or ring["dev"] != self.dev:
                ring = self.field.__dict__["_fused_ring"] = dict(
                    n=n, dev=self.dev, inbuf=[torch.empty(n, device=self.dev) for _ in range(_RING)],
                    pinned=[torch.empty(n, pin_memory=True) for _ in range(_RING)])
            self.inbuf[i], self.pinned[i], self._pin_np[i] = ring["inbuf"][i], ring["pinned"][i], None

    def _submit(self, i, m, graph, all_reduce=None):
        f = self.field
        cap = self.rows_cap
        if self.cap_once:                   # (tests: ONE submission with this capacity, whatever the records say)
            cap, self.cap_once = int(self.cap_once) // 32 * 32, 0
        if all_reduce is None:
            a, key = self._args(i, m["R"], m["stride"], m["N"], m["flags"], 3, cap, m["betas"], m["eps"], m["w_depth"], m["w_trans"], m["delta"])
            self._launch(i, a, key, graph, m)
        else:
            a, key = self._args(i, m["R"], m["stride"], m["N"], m["flags"], 1, cap, m["betas"], m["eps"], m["w_depth"], m["w_trans"], m["delta"])
            # parallel.ShardedExchange: the sharded optimiser — phase 1 seeds the TV gradient for the blocks this rank owns, the exchange
            # averages (reduce-scatter of the plane bodies + one small all-reduce), phase 2 steps the owned and the replicated blocks,
            # the exchange all-gathers the channel-last parameter bodies, phase 4 writes the gathered blocks to the caller's tensors
            sharded = hasattr(all_reduce, "gather") and int(getattr(all_reduce, "world", 1)) > 1
            a.shard_world, a.shard_rank = (int(all_reduce.world), int(all_reduce.rank)) if sharded else (0, 0)
            self._launch(i, a, key, False)
            f._gbuf_dirty, f._gbuf_stale, f._gbuf_reduced = True, False, False
            f._deferred_grad_key = f._uploaded_key
            self._install_head_grads()
            if hasattr(all_reduce, "reduce"):
                all_reduce.reduce(self)
            else:
                all_reduce()
            a.phases = 2
            self._launch(i, a, key, False)
            if sharded:
                all_reduce.gather(self)
                a.phases = 4
                self._launch(i, a, key, False)
            a.phases = 1
        self.slots[i] = dict(i=i, seq=self.issued, cap=cap, meta=m)
        self.issued += 1
        self._after_update()

    def step(self, rays, rgb_t, dep_t, N, flags, w_depth, w_trans, delta, tv, graph, all_reduce=None):
        f = self.field
        lib = _lib.load()
        untouched = self._key_after is not None and f._uploaded_key == self._key_after and \
            tuple((p.data_ptr(), p._version) for p in f._all_params()) == self._key_after
        h = f.sync_params()
        if self._handle_seen is not h:
            # a new native field (first step, or the handle was re-created): its step count is the optimiser's
            _, _, _, step0 = self._moments()
            with torch.cuda.device(self.dev):
                _lib.check(lib.t2n_field_train_set_step(h, step0, _lib.current_stream_ptr(self.dev)), "t2n_field_train_set_step")
            self._handle_seen = h
            torch.cuda.current_stream(self.dev).synchronize()      # (records of an earlier driver of this handle have landed)
            rec = (C.c_uint32 * 36)()
            _lib.check(lib.t2n_field_train_record(h, rec), "t2n_field_train_record")
            self.issued = self.seen = int(rec[0])
            self.slots = [None] * _RING
            self.queue = []
            self._drop_graphs()
        f.factor_grad_buffer(_raw=True)          # the field's gradient buffer exists and is the caller-owned one
        f._drop_preseed()
        self._poll()
        R, stride = int(rays.shape[0]), int(rays.shape[1])
        if not self.rows_cap:
            # nothing known yet: room for 12 appearance samples per ray (the first records correct it; a step that needs more is replayed)
            self.rows_cap = _ladder(12 * R)
        while True:
            i = self._free_slot()
            if not self.queue:
                break
            self._drain_queue()
        hyper, betas, eps = self._hyper(tv)
        n_in = R * stride + R + 3 * R + R + _lib.TRAIN_HYPER_FLOATS
        for j in range(_RING):          # (all four slots at once: a pinned allocation inside somebody's timed loop costs a millisecond)
            self._ensure_inbuf(j, n_in)
        buf, pin = self.inbuf[i], self.pinned[i]
        jitter = torch.rand(R, 1)      # CPU default generator, one draw per ray: models/tensorBase.py:313-317
        if self._pin_np[i] is None or self._pin_np[i].shape[0] != n_in:
            self._pin_np[i] = pin.numpy()
        o_h = n_in - _lib.TRAIN_HYPER_FLOATS
        self._pin_np[i][o_h:] = hyper
        pieces = [(rays, R * stride), (jitter, R), (rgb_t, 3 * R), (dep_t, R)]
        o = 0
        on_host = all(t.device.type == "cpu" for t, _ in pieces)
        if not on_host:
            buf[o_h:].copy_(pin[o_h:], non_blocking=True)
        for t, n in pieces:
            if t.numel() != n:
                raise T2NError(f"fused train step: batch tensor of {t.numel()} elements where {n} are expected")
            if t.device.type == "cpu":
                pin[o:o + n].copy_(t.reshape(-1))
                if not on_host:
                    buf[o:o + n].copy_(pin[o:o + n], non_blocking=True)
            else:
                buf[o:o + n].copy_(t.reshape(-1).float(), non_blocking=True)
            o += n
        # all on the host: the C call copies the staged batch itself — on its side stream, beside the previous step's tail, when nobody has
        # touched the field since this driver's last step (pipelined form, include/t2n.h); else in front of the step on the current stream
        self.host_batch = (C.c_void_p(pin.data_ptr()), n_in * 4) if on_host else None
        self.pipe_ok = bool(on_host and self.pipeline and untouched and not graph and all_reduce is None)
        if on_host and (graph or all_reduce is not None or not self.pipeline):
            buf.copy_(pin, non_blocking=True)
            self.host_batch = None
        meta = dict(R=R, stride=stride, N=N, flags=flags, betas=betas, eps=eps, w_depth=float(w_depth), w_trans=float(w_trans), delta=float(delta))
        self._submit(i, meta, graph, all_reduce)
        self.host_batch = None
        self._key_after = f._uploaded_key
        f.device_rows_steps = getattr(f, "device_rows_steps", 0) + 1
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        self.events.append(ev)
        self.slot_done[i] = ev
        if len(self.events) > _RUN_AHEAD:
            self.events.pop(0).synchronize()
        return self.losses

    def owns_head_grads(self, params):
        """True when `params` are exactly the 7 head tensors and their .grad are the views of this step's flat buffer."""
        ps = self.field._all_params()[12:]
        if len(params) != len(ps) or any(a is not b for a, b in zip(params, ps)):
            return False
        base = self.head_grads.data_ptr()
        off = 0
        for p, n in zip(ps, _lib.TRAIN_HEAD_SIZES):
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                return False
            off += n
        return True

    def _install_head_grads(self):
        """.grad of the 7 head tensors = views of the flat gradient buffer (+ the vote word rides behind them in an all-reduce)."""
        ps = self.field._all_params()[12:]
        off = 0
        for p, n in zip(ps, _lib.TRAIN_HEAD_SIZES):
            p.grad = self.head_grads[off:off + n].view_as(p)
            off += n
