"""hipGraph capture of a whole step (forward + backward) of the path.

At the reference's shipped batch size (`--batch-size 64` per GPU, lightning_module.py:468-473: ~25 k edges) a step of
the 4-layer network is ~700 kernel launches of a few microseconds each, and the host cannot issue them as fast as the
GPU retires them.  Every entry point of libcgat_hip launches on the caller's stream, allocates nothing and never
synchronises (include/cgat_hip.h), the layer's side stream forks from and joins the caller's stream through events, and
the CSR plans of a batch are cached -- so the whole step can be captured once into a hipGraph (torch.cuda.CUDAGraph is
the HIP graph API on ROCm) and replayed with ONE launch.

    step = GraphedStep(lambda: run(model, static_inputs))      # eager warm-up runs, then the capture
    out = step.replay()                                          # same buffers every time: copy new inputs into the
                                                                 # static input tensors before replaying

What a captured step may contain: anything built from this package's ops on tensors whose shapes do not change, i.e.
forward + backward of a layer or of CGAtNet on a batch of fixed shape (gradients land in the .grad tensors allocated
during the capture).  What it may not: index validation (a host round trip: `set_validate_indices(False)` or plans
already cached by the warm-up), the optimiser step (its bias corrections are host scalars), device collation of a
ragged batch (shapes change).  Outputs of EARLIER eager runs of the same step must not be alive when the capture starts
(drop them or `.detach()` them): their autograd graph pins AccumulateGrad nodes to the stream the eager run used."""
import torch

from . import ops


class GraphedStep:
    def __init__(self, fn, warmup=3, device=None):
        dev = torch.device(device if device is not None else torch.cuda.current_device())
        self.fn = fn
        import gc
        gc.collect()                                       # autograd graphs of earlier eager runs (see the module docstring)
        with torch.cuda.device(dev):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                  # warm-up off the default stream (as torch's capture recipe):
                for _ in range(max(1, warmup)):            # builds and caches the plans, sizes every workspace
                    fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            n0 = ops.prof_launches()
            with torch.cuda.graph(self.graph):
                self.out = fn()
            self.kernel_launches = ops.prof_launches() - n0    # library launches inside ONE step (torch's own not counted)

    def replay(self):
        self.graph.replay()
        return self.out
