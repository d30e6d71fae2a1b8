"""Whole-step hipGraph capture (MI355X: a 64^3 training step is ~300 kernel launches of ~30 us each, i.e. launch-bound on
the host; replaying one captured graph removes the per-launch host cost).

The step function (forward + loss + backward + clip + optimizer, as direct_regression/train_direct_4gpu.py:62-75 of the
reference) is captured once on static input buffers.  Dropout masks must still change from step to step although a replayed
graph passes identical kernel arguments, so every mask-drawing kernel offsets its seed by a device-resident step counter
(include/hvc_hip.h: hvc_set_seed_counter) that the first node of the graph advances.
"""
import torch

from . import _lib


class GraphedStep:
    """graphed = GraphedStep(step_fn, example_inputs); loss = graphed(*inputs) replays the captured step.

    step_fn(*inputs) must be capture-safe: static shapes, no host synchronisation (.item(), printing tensors), optimizer
    created with capturable=True.  It returns a tensor or a tuple / dict of tensors (e.g. the loss), which stay valid until the
    next call."""

    def __init__(self, step_fn, example_inputs, warmup=3, stream=None):
        """stream: the side stream to warm up AND capture on (default: a fresh one).  Under DistributedDataParallel pass the stream
        the DDP wrapper was constructed on: DDP hooks into AccumulateGrad nodes that live on its constructor's stream, and a
        capture taken on any other stream replays wrongly on ROCm 7 (measured, scripts/ddp_graph_probe2.py: every gradient
        non-finite after the first replay, although the eager steps on that stream pair are fine) - with constructor, warm-up
        and capture on ONE stream the captured DDP step, RCCL all-reduce included, replays bit for bit like the plain one."""
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs the MI355X HIP device")
        if warmup < 1:
            # lazily created state (AdamW's moments, cached bf16 weight copies) must exist BEFORE the capture: initialisations
            # recorded inside the graph would run again on every replay and reset that state
            raise ValueError("GraphedStep needs at least one eager warm-up step ahead of the capture")
        self.lib = _lib.load()
        self._closed = False
        self.static_inputs = [t.clone() for t in example_inputs]
        dev = self.static_inputs[0].device
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)          # read as uint32 on the device
        _lib.check(self.lib.hvc_set_seed_counter(self.counter.data_ptr()), "hvc_set_seed_counter")
        side = stream if stream is not None else torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._advance(side)
                step_fn(*self.static_inputs)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self._advance(side)
            self.static_outputs = step_fn(*self.static_inputs)
        # The graph has baked in the addresses of the cached compute-dtype / conv-layout weight copies (hvc.functional): hold
        # them, and refuse to replay once a parameter's copy has been rebuilt elsewhere (invalidate_param_casts, a
        # load_state_dict, a dtype switch) - the replay would read and refresh buffers nobody else uses any more.
        from . import functional as F
        self._pinned = []
        for (pid, attr), (ref, _) in list(F._CAST_REGISTRY.items()):
            prm = ref()
            hit = getattr(prm, attr, None) if prm is not None else None
            if hit is not None:
                self._pinned.append((ref, attr, hit[1]))

    def _advance(self, stream):
        _lib.check(self.lib.hvc_seed_counter_advance(self.counter.data_ptr(), 1, stream.cuda_stream), "hvc_seed_counter_advance")

    def __call__(self, *inputs):
        if self._closed:
            raise RuntimeError("GraphedStep was closed")
        from . import functional as F
        for ref, attr, dst in self._pinned:
            prm = ref()
            hit = getattr(prm, attr, None) if prm is not None else None
            if hit is None or hit[1] is not dst:
                raise RuntimeError("a cached weight copy captured by this graph was rebuilt (parameters re-cast or reloaded "
                                   "since the capture): build a new GraphedStep")
            # Same buffer, but is its CONTENT still the parameter's?  load_state_dict / copy_ into the parameter bumps its
            # version, invalidate_param_casts() the epoch: without an eager forward in between nothing has re-cast the copy, and
            # the captured forward would read pre-load weights for one whole step (the in-graph refresh runs after the
            # backward).  The graph only knows the buffer's address, so refill it in place and re-key it.
            key = hit[0]
            if key[:4] != (prm._version, F._CAST_EPOCH, prm.device, prm.data_ptr()):
                if prm.device != dst.device:
                    raise RuntimeError("a parameter captured by this graph moved to another device: build a new GraphedStep")
                with torch.no_grad():
                    if attr == "_hvc_cast":
                        dst.copy_(prm.detach())
                    else:
                        F._CONV_FILL[attr](dst, prm)
                setattr(prm, attr, (F._cache_key(prm, *key[4:]), dst))
                F._register(prm, attr)
        for dst, src in zip(self.static_inputs, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_outputs

    def close(self):
        """Detach the device counter from the library (seeds are taken as passed again).  Idempotent; also runs when the object
        is dropped or leaves a `with` block - the library must not keep reading a freed counter word."""
        if getattr(self, "_closed", True):      # never opened (the constructor failed early) or closed already
            return
        self._closed = True
        # only if the library still reads THIS object's counter: a newer GraphedStep may have installed its own since
        _lib.check(self.lib.hvc_clear_seed_counter_if(self.counter.data_ptr()), "hvc_clear_seed_counter_if")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
