"""Data-parallel gradient averaging for a static training step.

The reference wraps its model in torch DistributedDataParallel (direct_regression/train_direct_4gpu.py:146,
progressive_cascade/train_progressive_4gpu.py): gradients are averaged over the ranks, bucket by bucket, while the backward
pass is still running.  DDP drives that from one autograd hook PER PARAMETER plus per-iteration bookkeeping; on this model that
is ~19 ms of host time per 128^3 step (bench.py `host_enqueue_ms_per_step`: 33 ms plain, 52 ms under DDP, whatever its
bucket size / static_graph / broadcast_buffers settings) against a 54 ms GPU step - the launch thread no longer runs ahead of the
GPU and every rank of an N > 1 job loses 5 - 9 % to the host, before a byte moves over xGMI.

BucketedGradReducer keeps DDP's contract and its overlap, with the host work of a handful of hooks:
  * all gradients end up in ONE flat fp32 buffer laid out in the order in which the backward pass produces them, so a bucket is a
    contiguous slice; the backward pass itself hands every parameter a fresh gradient tensor (no per-parameter accumulate kernel, as
    bucket views cost), and the bucket's hook gathers its members with one multi-tensor copy and points their `.grad` at the slots;
  * the first backward (per-parameter hooks, once) records that order; from then on only the LAST parameter of each bucket
    carries a hook, which all-reduces the bucket's slice on a side stream (RCCL over xGMI; AVG);
  * finish() - after backward, before clipping / optimizer.step - joins the side stream.
The step must be static (same parameters receive gradients in the same order every step), which the reference's training loops
are; `check=True` keeps the per-parameter hooks and asserts the order on every step (tests).
Use reducer.zero_grad() instead of optimizer.zero_grad(): it drops the gradients (set-to-none semantics).
"""
import torch
import torch.distributed as dist


class BucketedGradReducer:
    def __init__(self, params, group=None, bucket_bytes=32 << 20, check=False):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("BucketedGradReducer: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("BucketedGradReducer: parameters must share one device and dtype")
        self.group, self.bucket_bytes, self.check = group, int(bucket_bytes), bool(check)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.device = dev
        self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._order, self._hooks, self._work, self._launched = [], [], [], set()
        self._buckets = None                    # [(start, end, [member parameter indices])] once the arrival order is known
        self._fired = set()
        self._layout(list(range(len(self.params))))
        for i, p in enumerate(self.params):     # discovery: one backward with a hook on every parameter
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_discovery_hook(i)))

    # ---- layout ---------------------------------------------------------------------------------------------------------------
    def _layout(self, order):
        """Flat buffer holding the gradients of the parameters in `order` (zeros until a step writes them)."""
        self.order = order
        sizes = [self.params[i].numel() for i in order]
        self.flat = torch.zeros(sum(sizes), dtype=self.params[0].dtype, device=self.device)
        self.offsets, self.views, off = {}, {}, 0
        for i, n in zip(order, sizes):
            self.offsets[i] = (off, off + n)
            self.views[i] = self.flat[off:off + n].view_as(self.params[i])
            off += n

    def _make_discovery_hook(self, i):
        def hook(_p):
            self._order.append(i)
        return hook

    def _make_bucket_hook(self, b):
        def hook(_p):
            if self.check:
                missing = [i for i in self._buckets[b][2] if i not in self._fired]
                if missing:
                    raise RuntimeError(f"BucketedGradReducer: bucket {b} closed before the gradients of parameters {missing} arrived "
                                       "(the step is not static)")
            self._launch(b)
        return hook

    def _rebuild(self):
        """After the discovery backward: lay the buffer out in arrival order, cut it into buckets, keep one hook per bucket.
        Parameters that received no gradient get no slot: their .grad stays None, as without any reducer."""
        seen = list(dict.fromkeys(self._order))
        if self.world > 1:                      # every rank must cut the same buckets: compare the arrival orders once (fail loudly, not silently)
            mine = torch.tensor(seen + [-1] * (len(self.params) - len(seen)), dtype=torch.int64, device=self.device)
            lo, hi = mine.clone(), mine.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if not torch.equal(lo, hi):
                raise RuntimeError("BucketedGradReducer: the ranks' backward passes produced their gradients in different orders")
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._layout(seen)
        esize = self.flat.element_size()
        buckets, start, members = [], 0, []
        for i in seen:
            members.append(i)
            end = self.offsets[i][1]
            if (end - start) * esize >= self.bucket_bytes:
                buckets.append((start, end, members))
                start, members = end, []
        if members:
            buckets.append((start, self.flat.numel(), members))
        self._buckets = buckets
        if self.check:                          # (registered first: a parameter's hooks run in registration order)
            for i in seen:
                self._hooks.append(self.params[i].register_post_accumulate_grad_hook(lambda _p, i=i: self._fired.add(i)))
        for b, (_, _, mem) in enumerate(buckets):   # the bucket closes with the gradient that arrives last in it
            self._hooks.append(self.params[mem[-1]].register_post_accumulate_grad_hook(self._make_bucket_hook(b)))

    # ---- per step -------------------------------------------------------------------------------------------------------------
    def zero_grad(self):
        """Call instead of optimizer.zero_grad(): the gradients are dropped (the backward pass hands each parameter a fresh tensor, no
        accumulate kernel), and each bucket's hook gathers them into its slice of the flat buffer, which p.grad then views."""
        if self._buckets is None and self._order:
            self._rebuild()                     # the discovery step is over: its gradients have been consumed
        self._order, self._fired, self._launched = [], set(), set()
        for p in self.params:
            p.grad = None

    def _gather(self, members):
        """The members' freshly produced gradients -> their slots (one multi-tensor copy); p.grad become the slots."""
        src, dst = [], []
        for i in members:
            g = self.params[i].grad
            if g is None:
                self.views[i].zero_()           # no gradient this step (possible only when the step is not static)
            elif g.data_ptr() != self.views[i].data_ptr():
                src.append(g)
                dst.append(self.views[i])
            self.params[i].grad = self.views[i]
        if src:
            torch._foreach_copy_(dst, src)

    def _all_reduce(self, view):
        if self.backend == "nccl":
            return dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)      # gloo (tests): no AVG
        return (work, view)

    def _launch(self, b):
        start, end, members = self._buckets[b]
        self._launched.add(b)
        self._gather(members)
        view = self.flat[start:end]
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                self._work.append(self._all_reduce(view))
        else:
            self._work.append(self._all_reduce(view))

    def finish(self):
        """After backward: reduce what no hook has (the discovery step: everything) and make the gradients visible to the caller's stream."""
        if self._buckets is None:
            self._buckets = [(0, self.flat.numel(), list(self.order))]
            self._launch(0)
            self._buckets = None
        else:
            for b in range(len(self._buckets)):
                if b not in self._launched:
                    self._launch(b)
        for w in self._work:
            if isinstance(w, tuple):
                w[0].wait()
                w[1].div_(self.world)
            else:
                w.wait()
        self._work = []
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        if self._buckets is None:               # discovery step: parameters that received nothing keep .grad = None
            got = set(self._order)
            for i, p in enumerate(self.params):
                if i not in got:
                    p.grad = None

    def describe(self):
        nb = len(self._buckets) if self._buckets else 0
        return {"reducer": "BucketedGradReducer (flat fp32 gradients in arrival order, one autograd hook per bucket, all-reduce AVG on a side stream)",
                "buckets": nb, "bucket_cap_mb": self.bucket_bytes / 2 ** 20, "parameters": len(self.params),
                "gradient_mb": self.flat.numel() * self.flat.element_size() / 2 ** 20}


def broadcast_module_state(module, src=0, group=None):
    """Rank `src`'s parameters and buffers to every rank (what DistributedDataParallel's constructor does), coalesced per dtype."""
    with torch.no_grad():
        tensors = [t for t in list(module.parameters()) + list(module.buffers()) if t.numel()]
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault((t.dtype, t.device), []).append(t)
        for (_dt, _dev), ts in by_dtype.items():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            dist.broadcast(flat, src, group=group)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
