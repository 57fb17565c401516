"""FitPipeline: throughput mode for a STREAM of same-shaped batches.

One launch of the fit kernel ends with a straggler tail: a few fits need >100 LM iterations while most of the GPU
is already idle.  Consecutive batches are independent, so the pipeline keeps ``n_slots`` handles,
each on its own HIP stream, and submits batch k to slot k mod n_slots: the tail of one launch overlaps the bulk of the
next.  Every batch is fitted completely and independently; only the submission is overlapped.  Results come back as
torch tensors that are valid once the slot's stream has reached them (``wait()`` or stream semantics).

ROCm maps HIP streams onto 4 hardware queues by default; with more than two slots (or next to RCCL's stream) set
``GPU_MAX_HW_QUEUES=8`` in the environment before the first HIP call, or streams sharing a queue serialise.
"""
from .batch import BatchProblem

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


class FitPipeline:
    def __init__(self, model, first_batch, x=None, weights=None, epsilon=None, n_slots=2):
        """first_batch: a torch CUDA tensor (B, m) or (B, S, m) that fixes the batch shape and the device"""
        if torch is None or not isinstance(first_batch, torch.Tensor) or not first_batch.is_cuda:
            raise ValueError("FitPipeline works on torch CUDA tensors (device-pointer mode)")
        self.device = first_batch.device
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(int(n_slots))]
        self.slots = []
        for st in self.streams:
            st.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(st):
                self.slots.append(BatchProblem(model, first_batch, x=x, weights=weights, epsilon=epsilon))
        self._k = 0

    def submit(self, Y, alpha0, solver=None, want_coefficients=True):
        """enqueue the fit of one batch (asynchronous); returns (alpha, C, report, slot_index).  Y and alpha0 must be
        ready on the CURRENT stream (the slot's stream waits for it); they may be dropped right after the call.
        The results are valid on the current stream after wait(slot)."""
        i = self._k % len(self.slots)
        self._k += 1
        st = self.streams[i]
        cur = torch.cuda.current_stream(self.device)
        st.wait_stream(cur)
        # Y / alpha0 were allocated on the caller's stream and are read on the slot's stream: tell the caching
        # allocator, or a caller that drops them right after submit() (`pipe.submit(Y.cuda(), g)` in a loop) has
        # their memory handed to the next H2D copy while the slot is still reading it
        for t in (Y, alpha0):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(st)
        with torch.cuda.stream(st):
            h = self.slots[i]
            h.set_observations(Y)
            alpha, C, rep = h.fit(alpha0, solver=solver, want_coefficients=want_coefficients)
        # the results are allocated on the slot's stream and consumed on the caller's stream after wait()
        for t in (alpha, C, rep):
            if isinstance(t, torch.Tensor):
                t.record_stream(cur)
        return alpha, C, rep, i

    def wait(self, slot=None):
        """make the current stream wait for one slot (or all of them): results are then safe to use on it"""
        cur = torch.cuda.current_stream(self.device)
        for i, st in enumerate(self.streams):
            if slot is None or slot == i:
                cur.wait_stream(st)

    def report_to_numpy(self, rep):
        return BatchProblem.report_to_numpy(rep)

    def close(self):
        for h in self.slots:
            h.close()
        self.slots = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
