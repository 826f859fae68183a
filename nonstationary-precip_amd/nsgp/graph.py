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


class GraphedSequence:
    """fns[0], between[0], fns[1], between[1], ...: every `fn` captured as its own hipGraph, the `between` callables
    (collectives of the data-parallel step) stay eager calls between the replays.  The graphs share one memory pool --
    tensors made by one part (autograd graph, gradients of the cut leaves of a staged backward) are consumed by the
    next -- and are always replayed in capture order."""

    def __init__(self, fns, between, warmup=3):
        assert len(between) == len(fns)
        self.fns, self.between = list(fns), list(between)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                for fn, bt in zip(self.fns, self.between):
                    fn()
                    if bt is not None:
                        bt()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        self.graphs, self.outs = [], []
        for fn, bt in zip(self.fns, self.between):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode='thread_local'):
                out = fn()
            self.graphs.append(g)
            self.outs.append(out)
            if bt is not None:
                bt()

    def __call__(self):
        for g, bt in zip(self.graphs, self.between):
            g.replay()
            if bt is not None:
                bt()
        return self.outs
