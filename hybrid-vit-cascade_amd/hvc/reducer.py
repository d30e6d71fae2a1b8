"""Data-parallel gradient averaging for a static training step.

The reference wraps its model in torch DistributedDataParallel (direct_regression/train_direct_4gpu.py:146,
progressive_cascade/train_progressive_4gpu.py): gradients are averaged over the ranks, bucket by bucket, while the backward
pass is still running.  DDP drives that from one autograd hook PER PARAMETER plus per-iteration bookkeeping; on this model that
is ~19 ms of host time per 128^3 step (bench.py `host_enqueue_ms_per_step`: 33 ms plain, 52 ms under DDP, whatever its
bucket size / static_graph / broadcast_buffers settings) against a 54 ms GPU step - the launch thread no longer runs ahead of the
GPU and every rank of an N > 1 job loses 5 - 9 % to the host, before a byte moves over xGMI.

BucketedGradReducer keeps DDP's contract and its overlap, with the host work of a handful of hooks:
  * all gradients live in ONE flat fp32 buffer (`p.grad` are views, like DDP's gradient_as_bucket_view), laid out in the order
    in which the backward pass produces them, so a bucket is a contiguous slice;
  * the first backward (per-parameter hooks, once) records that order; from then on only the LAST parameter of each bucket
    carries a hook, which all-reduces the bucket's slice on a side stream (RCCL over xGMI; AVG);
  * finish() - after backward, before clipping / optimizer.step - joins the side stream.
The step must be static (same parameters receive gradients in the same order every step), which the reference's training loops
are; `check=True` keeps the per-parameter hooks and asserts the order on every step (tests).
Gradients must stay allocated: use reducer.zero_grad() instead of optimizer.zero_grad(set_to_none=True).
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
        self._buckets = None                    # [(start, end, sentinel parameter index)] once the order is known
        self._fired = set()
        self._layout(list(range(len(self.params))))
        for i, p in enumerate(self.params):     # discovery: one backward with a hook on every parameter
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_discovery_hook(i)))

    # ---- layout ---------------------------------------------------------------------------------------------------------------
    def _layout(self, order):
        """Flat buffer with the parameters' gradients in `order`; p.grad become views of it (contents are not carried over)."""
        self.order = order
        sizes = [self.params[i].numel() for i in order]
        self.flat = torch.zeros(sum(sizes), dtype=self.params[0].dtype, device=self.device)
        self.offsets, off = {}, 0
        for i, n in zip(order, sizes):
            self.offsets[i] = (off, off + n)
            self.params[i].grad = self.flat[off:off + n].view_as(self.params[i])
            off += n

    def _make_discovery_hook(self, i):
        def hook(_p):
            self._order.append(i)
        return hook

    def _make_bucket_hook(self, b):
        def hook(_p):
            if self.check:
                start, end, _ = self._buckets[b]
                missing = [i for i in self.order if start <= self.offsets[i][0] < end and i not in self._fired]
                if missing:
                    raise RuntimeError(f"BucketedGradReducer: bucket {b} closed before the gradients of parameters {missing} arrived "
                                       "(the step is not static)")
            self._launch(b)
        return hook

    def _rebuild(self):
        """After the discovery backward: lay the buffer out in arrival order, cut it into buckets, keep one hook per bucket."""
        seen = list(dict.fromkeys(self._order))
        rest = [i for i in range(len(self.params)) if i not in set(seen)]       # parameters that received no gradient: at the end
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._layout(seen + rest)
        esize = self.flat.element_size()
        buckets, start, last = [], 0, None
        for i in seen:
            last = i
            end = self.offsets[i][1]
            if (end - start) * esize >= self.bucket_bytes:
                buckets.append((start, end, i))
                start, last = end, None
        total = self.flat.numel()
        if start < total:
            # the tail (incl. parameters without gradients) closes with the last parameter that does arrive - or, if none is left
            # in it, it is reduced by finish()
            buckets.append((start, total, last))
        self._buckets = buckets
        if self.check:                          # (registered first: a parameter's hooks run in registration order)
            for i in seen:
                self._hooks.append(self.params[i].register_post_accumulate_grad_hook(lambda _p, i=i: self._fired.add(i)))
        for b, (_, _, sentinel) in enumerate(buckets):
            if sentinel is not None:
                self._hooks.append(self.params[sentinel].register_post_accumulate_grad_hook(self._make_bucket_hook(b)))

    # ---- per step -------------------------------------------------------------------------------------------------------------
    def zero_grad(self):
        """Call instead of optimizer.zero_grad(): gradients stay allocated (views of the flat buffer)."""
        if self._buckets is None and self._order:
            self._rebuild()                     # the discovery step is over: its gradients have been consumed
        self._order, self._fired, self._launched = [], set(), set()
        self.flat.zero_()
        for i in self.order:                    # an optimizer or a user may have dropped them (set_to_none): re-attach
            p = self.params[i]
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + self.offsets[i][0] * self.flat.element_size():
                a, b = self.offsets[i]
                p.grad = self.flat[a:b].view_as(p)

    def _all_reduce(self, view):
        if self.backend == "nccl":
            return dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)      # gloo (tests): no AVG
        return (work, view)

    def _launch(self, b):
        start, end, _ = self._buckets[b]
        self._launched.add(b)
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
            self._launch_all()
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

    def _launch_all(self):
        self._buckets = [(0, self.flat.numel(), None)]
        self._launch(0)
        self._buckets = None

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
