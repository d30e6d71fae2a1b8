"""Host-to-device prefetch for the trainers' batch loops (reference train_direct_4gpu.py:62-66 copies each batch with
`.cuda(rank, non_blocking=True)` right before the step, so the copy of a 2 x 128^3 target (16 MB) + two 512^2 views waits in
front of the step's first kernel).  DevicePrefetcher keeps ONE batch ahead on a side HIP stream: while step t runs on the
compute stream, batch t+1 is copied from pinned host memory; the compute stream then only waits on an event.  Same items, same
order, same dict keys as the wrapped loader."""
import torch


class DevicePrefetcher:
    def __init__(self, loader, device, keys=("drr_stacked", "ct_volume")):
        self.loader, self.device, self.keys = loader, torch.device(device), tuple(keys)
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher copies to the MI355X HIP device only")
        self.stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        staged = dict(batch)
        with torch.cuda.stream(self.stream):
            for k in self.keys:
                if k in batch and torch.is_tensor(batch[k]):
                    staged[k] = batch[k].to(self.device, non_blocking=True)
        return staged

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = nxt
            torch.cuda.current_stream(self.device).wait_stream(self.stream)      # batch t is complete before the step uses it
            for k in self.keys:
                if k in cur and torch.is_tensor(cur[k]):
                    cur[k].record_stream(torch.cuda.current_stream(self.device))   # its memory outlives the side stream's use
            try:
                nxt = self._stage(next(it))                                       # batch t+1 copies while step t runs
            except StopIteration:
                nxt = None
            yield cur
