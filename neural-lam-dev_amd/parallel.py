"""Pure data parallelism for the hot path: one process per GPU, the full graph
and model replicated, samples sharded, ONE exchange per step -- the gradient
all-reduce (the reference's Lightning `strategy="ddp"`, train_model.py:279).

MI355X-first choices:
  * parameters live in one flat fp32 buffer (FlatParams); gradients are packed
    into one flat buffer, so the exchange is a single RCCL all-reduce (GraphLAM:
    0.86 MB, latency-bound) or a few large buckets (Hi-LAM: 22-89 MB) rather
    than per-parameter messages; xGMI is point-to-point, so fewer and larger
    collectives win;
  * buckets are reduced asynchronously in reverse parameter order while the
    packing of earlier buckets proceeds;
  * the optimiser is one fused AdamW kernel over the flat buffers
    (nlam_adamw_step), with the 1/world scale folded in.
Works unchanged on CPU with the gloo backend for the collective part (tests).
"""
import torch
import torch.distributed as dist


class FlatParams:
    """Re-homes every parameter of `module` into one contiguous fp32 buffer
    (views keep names/shapes, so state_dict is unchanged).  Every parameter starts on
    a 64-byte boundary (the kernels' float4 weight loaders need 16-byte aligned rows; a
    17-wide bias would otherwise misalign everything behind it); the padding stays zero."""

    ALIGN = 16   # elements

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        dev = self.params[0].device
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += -(-p.numel() // self.ALIGN) * self.ALIGN
        self.numel = off
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.flat[o : o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o : o + n].view(p.shape)
        self.grad = torch.zeros_like(self.flat)
        self._pad = {}   # pad length -> zeros
        self._tables = {}     # (lo, hi) -> last pack table of that range (_pack_native)
        self._captured = []   # tables a HIP-graph capture refers to
        self._spare = {}      # (lo, hi) -> page-locked buffer put aside for a later capture

    def span(self, lo, hi):
        """[a, b) element range of params[lo:hi] in the flat buffers (padding included)."""
        a = self.offsets[lo]
        b = self.offsets[hi] if hi < len(self.params) else self.numel
        return a, b

    def gather(self, buf):
        """The parameters' elements of a flat buffer, padding removed."""
        return torch.cat([buf[o : o + p.numel()] for p, o in zip(self.params, self.offsets)])

    def _zeros(self, n):
        z = self._pad.get(n)
        if z is None:
            z = self._pad[n] = torch.zeros(n, dtype=torch.float32, device=self.flat.device)
        return z

    def pack_grads(self, lo=0, hi=None):
        """Copy p.grad of params[lo:hi] into their slice of the flat grad buffer."""
        hi = len(self.params) if hi is None else hi
        if hi <= lo:
            return
        a, b = self.span(lo, hi)
        if self.grad.is_cuda:
            return self._pack_native(lo, hi)
        parts = []
        for i in range(lo, hi):
            p = self.params[i]
            n = p.numel()
            parts.append(p.grad.reshape(-1) if p.grad is not None else self._zeros(n))
            end = self.offsets[i + 1] if i + 1 < len(self.params) else self.numel
            pad = end - self.offsets[i] - n
            if pad:
                parts.append(self._zeros(pad))
        torch.cat(parts, out=self.grad[a:b])

    def _pack_native(self, lo, hi):
        """One launch (nlam_pack_segments) for the whole range.  The table of source addresses is
        uploaded only when it differs from the one last used for this range: under HIP-graph
        replay it never does, and in eager steps the caching allocator hands backward the same
        blocks step after step.  Tables made during a graph capture are kept for the life of
        this object (replays read the captured host copy again)."""
        from . import ops
        from ._lib import lib

        srcs = []
        for i in range(lo, hi):
            g = self.params[i].grad
            if g is not None and not (g.is_contiguous() and g.dtype == torch.float32):
                g = g.contiguous().float()
                self.params[i].grad = g
            srcs.append(g.data_ptr() if g is not None else 0)
        key = (lo, hi, tuple(srcs))
        hit = self._tables.get((lo, hi))
        if hit is None or hit[0] != key:
            chunk = int(lib.nlam_pack_chunk())
            first, rows = [0], []
            for i, ptr in zip(range(lo, hi), srcs):
                n = self.params[i].numel()
                rows += [ptr, self.offsets[i], n]
                first.append(first[-1] + -(-n // chunk))
            table = torch.tensor(rows + first, dtype=torch.int64)
            if torch.cuda.is_current_stream_capturing():
                # page-locked memory cannot be allocated inside a capture: take the buffer an
                # earlier eager call of this range put aside; the capture owns it from here on
                host = self._spare.pop((lo, hi), None)
                if host is None:
                    raise RuntimeError("FlatParams.pack_grads: run one eager step before capturing "
                                       "it into a HIP graph (the address table needs a page-locked "
                                       "buffer allocated outside the capture)")
                host.copy_(table)
            else:
                host = table.pin_memory()
            dev = torch.empty(host.numel(), dtype=torch.int64, device=self.grad.device)
            dev.copy_(host, non_blocking=True)
            capturing = torch.cuda.is_current_stream_capturing()
            ev = None
            if not capturing:
                # a later cache hit may launch on another stream (e.g. warm-up on a side stream,
                # then eager steps on the main one): that launch waits for this upload
                ev = torch.cuda.Event()
                ev.record()
            hit = (key, dev, first[-1], host, ev, torch.cuda.current_stream().cuda_stream)
            if capturing:
                self._captured.append(hit)
            self._tables[(lo, hi)] = hit
        if not torch.cuda.is_current_stream_capturing() and (lo, hi) not in self._spare:
            # (every eager call, hit or miss: a capture takes the spare of its range)
            self._spare[(lo, hi)] = torch.empty(3 * (hi - lo) + (hi - lo + 1),
                                                dtype=torch.int64).pin_memory()
        _, dev, nchunks, _, ev, up_stream = hit
        if ev is not None and up_stream != torch.cuda.current_stream().cuda_stream:
            torch.cuda.current_stream().wait_event(ev)
        ops._launch("nlam_pack_segments", lib.nlam_pack_segments,
                    (dev.data_ptr(), hi - lo, nchunks, self.grad.data_ptr(), ops.stream()))

    def zero_grad(self):
        for p in self.params:
            p.grad = None


def bucket_ranges(sizes, bucket_numel):
    """Split parameter indices [0, n) into contiguous ranges of about
    bucket_numel elements, returned in REVERSE order (last layers first, the
    order backward produces them)."""
    ranges, start, acc = [], 0, 0
    for i, s in enumerate(sizes):
        acc += s
        if acc >= bucket_numel:
            ranges.append((start, i + 1))
            start, acc = i + 1, 0
    if start < len(sizes):
        ranges.append((start, len(sizes)))
    return ranges[::-1]


class GradAllReduce:
    """Bucketed flat-buffer gradient all-reduce (SUM; the 1/world factor is
    applied by the optimiser).

    Payloads of at least `min_overlap_bytes` (Hi-LAM: 22-89 MB) are overlapped with
    backward the way Lightning's DDP does for the reference (train_model.py:279): every
    parameter carries a post-accumulate-grad hook; when the last gradient of a bucket has
    been produced, that bucket is packed and its all-reduce is issued on a side stream
    while autograd keeps going; reduce() only issues what is left (parameters that got no
    gradient) and waits.  Small payloads (GraphLAM: 0.86 MB, latency-bound) stay ONE
    collective after backward.  Bucket contents and summation are identical either way, so
    the two schedules give bit-identical results."""

    def __init__(self, flat, bucket_bytes=32 << 20, group=None, overlap=None,
                 min_overlap_bytes=4 << 20, single_rank_collectives=False):
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # single_rank_collectives: issue every collective even in a ONE-rank group -- the RCCL
        # calls, streams and hooks of the multi-GPU path executed on a one-GPU box (a one-rank
        # all-reduce returns its input: results are those of the plain path)
        self.active = self.world > 1 or (bool(single_rank_collectives) and dist.is_initialized())
        sizes = [p.numel() for p in flat.params]
        self.bucket_bytes = bucket_bytes
        self.ranges = bucket_ranges(sizes, max(1, bucket_bytes // 4))
        if overlap is None:
            overlap = self.active and flat.numel * 4 >= min_overlap_bytes
        self.overlap = bool(overlap)
        self.side = None
        self._handles = []
        self._launched = set()
        self._dirty = set()
        self.stats = {"launched_in_backward": 0, "launched_in_reduce": 0}
        # False while a HIP graph of the step is captured / replayed: the hooks must not issue
        # collectives then (reduce(packed=True) issues every bucket after the replay instead)
        self.hooks_enabled = True
        if self.overlap:
            from . import ops

            import weakref

            ops.OVERLAP_REDUCERS.append(weakref.ref(self))   # (ops.deferral_allowed)
            self._bucket_of = {}
            self._pending = []
            for bi, (lo, hi) in enumerate(self.ranges):
                self._pending.append(hi - lo)
                for i in range(lo, hi):
                    self._bucket_of[i] = bi
            self._left = list(self._pending)
            self._seen = [False] * len(flat.params)
            for i, p in enumerate(flat.params):
                p.register_post_accumulate_grad_hook(self._make_hook(i))

    def describe(self):
        return {"world": self.world, "buckets": len(self.ranges),
                "bucket_bytes": self.bucket_bytes, "payload_bytes": self.flat.numel * 4,
                "overlap_with_backward": self.overlap}

    def broadcast_params(self, src=0):
        if self.active:
            dist.broadcast(self.flat.flat, src=src, group=self.group)

    # ---- overlap machinery ---------------------------------------------------
    def _make_hook(self, i):
        def hook(_param):
            if not self.hooks_enabled:
                return
            bi = self._bucket_of[i]
            if bi in self._launched:
                # a second gradient contribution after the bucket went out (a re-entrant
                # inner backward, e.g. args.ar_checkpoint): re-reduce it in reduce()
                self._dirty.add(bi)
                return
            if self._seen[i]:
                return   # the same parameter again (re-entrant backward): counted once
            self._seen[i] = True
            self._left[bi] -= 1
            if self._left[bi] == 0 and self.active:
                self._launch(bi, in_backward=True)
        return hook

    def _launch(self, bi, in_backward, packed=False):
        f = self.flat
        lo, hi = self.ranges[bi]
        a, b = f.span(lo, hi)
        if f.grad.is_cuda and self.overlap:
            if self.side is None:
                self.side = torch.cuda.Stream(device=f.grad.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(f.grad.device))
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                if not packed:
                    f.pack_grads(lo, hi)
                h = dist.all_reduce(f.grad[a:b], op=dist.ReduceOp.SUM, group=self.group,
                                    async_op=True)
        else:
            if not packed:
                f.pack_grads(lo, hi)
            h = dist.all_reduce(f.grad[a:b], op=dist.ReduceOp.SUM, group=self.group,
                                async_op=True)
        self._handles.append(h)
        self._launched.add(bi)
        self.stats["launched_in_backward" if in_backward else "launched_in_reduce"] += 1

    def reduce(self, packed=False):
        """Pack (unless the caller already did) + all-reduce every bucket that backward's
        hooks have not issued yet; returns when all are complete on the current stream."""
        f = self.flat
        if not self.active:
            if not packed:
                f.pack_grads()
            self._reset()
            return
        for bi in range(len(self.ranges)):
            if bi not in self._launched:
                self._launch(bi, in_backward=False, packed=packed)
        for h in self._handles:
            h.wait()
        if self._dirty:
            self._handles = []
            for bi in sorted(self._dirty):
                self._launch(bi, in_backward=False)
            for h in self._handles:
                h.wait()
        if self.side is not None:
            torch.cuda.current_stream(f.grad.device).wait_stream(self.side)
        self._reset()

    def _reset(self):
        self._handles = []
        self._launched = set()
        self._dirty = set()
        if self.overlap:
            self._left = list(self._pending)
            self._seen = [False] * len(self._seen)


class FlatAdamW:
    """AdamW(lr, betas=(0.9, 0.95)) of ar_model.py:191-195 as one kernel over the
    flat parameter buffer."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01):
        self.flat, self.lr, self.betas, self.eps, self.wd = flat, lr, betas, eps, weight_decay
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.t = 0

    def step(self, grad_scale=1.0):
        from . import ops
        from ._lib import check, lib

        self.t += 1
        f = self.flat
        ops._require_dev(f.flat, "flat parameter buffer")
        check(
            lib.nlam_adamw_step(
                f.flat.data_ptr(), f.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                f.numel, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t,
                grad_scale, ops.stream(),
            ),
            "nlam_adamw_step",
        )


class GraphedTrainStep:
    """forward + loss + backward + gradient packing of one training step captured
    into a HIP graph (torch.cuda.CUDAGraph over the launch stream) and replayed:
    the step is ~150 short launches, so eager dispatch leaves gaps between kernels.
    The batch tensors are static (copy new data into them between replays); the
    collective and the AdamW kernel (whose bias correction depends on the step
    count) stay outside the graph.  Falls back to eager if capture is refused."""

    def __init__(self, model, flat, batch, warmup=3):
        self.model, self.flat, self.batch = model, flat, batch
        self.graph = None
        self.loss = None
        # d loss / d loss, made once (loss.backward() would fill a fresh ones tensor every step)
        self._one = torch.ones((), dtype=torch.float32, device=flat.flat.device)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._eager()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            flat.zero_grad()
            with torch.cuda.graph(g):
                loss = model.training_step(batch)
                loss.backward(self._one)
                flat.pack_grads()
            self.graph, self.loss = g, loss
        except Exception as e:  # pragma: no cover - depends on the runtime
            import warnings

            warnings.warn(f"HIP graph capture failed ({e!r}); running the step eagerly")
            self.graph = None
            torch.cuda.synchronize()

    def _eager(self):
        self.flat.zero_grad()
        loss = self.model.training_step(self.batch)
        loss.backward(self._one)
        self.flat.pack_grads()
        return loss

    def __call__(self):
        if self.graph is None:
            return self._eager()
        self.graph.replay()
        return self.loss
