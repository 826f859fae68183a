"""hipGraph capture of a training step (torch.cuda.CUDAGraph is hipGraph on ROCm).

A DSVI step is ~230 short launches; replaying them as one graph removes the per-launch host cost.
Everything inside must be replay-safe: static input buffers, no host synchronisation, step counters
on the device (FusedAdam(capturable=True), PhiloxEps(step_dev=...)).  All nsgp kernels are launched on
torch's current stream, which is the capturing stream inside `torch.cuda.graph`."""
import torch


class GraphedCallable:
    def __init__(self, fn, warmup=3):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                 # allocator / lazy-init warm-up must happen before capture
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: API calls of OTHER host threads (e.g. the RCCL watchdog polling events of earlier collectives)
        # must not invalidate this thread's capture
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out
