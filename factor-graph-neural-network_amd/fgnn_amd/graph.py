"""hipGraph capture of a launch-bound step.

One LDPCModel training step is ~1500 kernel launches of 5-250 us each; at the reference batch sizes the
Python / HIP launch path (~20 us per launch) costs as much as the kernels themselves.  ``StepGraph`` records
the step once on a capture stream (every kernel of this package launches on torch's current stream, so the
hand-written HIP kernels, rocBLAS GEMMs and ATen elementwise ops all land in the same graph) and replays it
with a single launch.  Requirements are torch.cuda.graphs' usual ones: static input tensors (copy new data
into them), no host synchronisation inside the step, and no collectives (keep the gradient all-reduce outside).
"""
import torch


class StepGraph:
    def __init__(self, fn, warmup=2, static_params=False):
        """``static_params=True`` (inference of a frozen model): the parameter-derived tensors — folded BatchNorm affines, bf16
        weight copies — are the ones the warm-up runs left in their caches, and the graph holds no kernels that rebuild them
        (the LDPC inference forward otherwise replays ~260 tiny fold / cast / copy kernels per step beside ~60 real ones).
        Replays then do NOT see parameter changes made after the capture: build a new StepGraph after loading other weights.
        Default (False): the caches are invalidated before the capture, so the refresh kernels are recorded and every replay
        derives them from the parameters as they are at that moment (what a training step needs).

        ``fn()`` is run ``warmup`` times (allocator / workspace / autotune warm-up), then captured — both on ONE side
        stream.  Autograd runs a leaf's gradient accumulation on the stream the leaf was first used on; if the warm-up ran
        on another stream than the capture, parameters whose gradients come through autograd (plain torch modules such as
        the edge models of train_syn_*.py, not this package's gradient-sink kernels) would be accumulated on a branch of
        the graph while the allocator, which only knows the capture stream, hands the incoming gradient's memory to the
        next kernel: replayed steps then add garbage into those gradients (tools/diag_graph.py)."""
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        # cached low-precision weight copies are refreshed in place where they are stale: make them stale now, so the
        # refresh kernels are captured and every replay casts the parameters as they are at that moment
        from .mpnn import pointwise
        self.static_params = bool(static_params)
        if not self.static_params:
            pointwise.invalidate_casts()
        self.graph = torch.cuda.CUDAGraph()
        import os
        dot = os.environ.get('FGNN_GRAPH_DOT')          # diagnosis: the captured graph (nodes, edges) as hipGraphDebugDotPrint writes it
        if dot:
            self.graph.enable_debug_mode()
        from . import ops
        self.stamps = None
        if os.environ.get('FGNN_STAMPS'):               # diagnosis: device timestamps inside the replayed step (ops.stamp)
            ops.STAMPS = {'buf': torch.zeros(8192, dtype=torch.int64, device='cuda'), 'tags': []}
        delay_ms = float(os.environ.get('FGNN_STEP_HEAD_START_MS', '0'))     # diagnosis (profiled runs): see fgnn_spin
        with torch.cuda.graph(self.graph, stream=self.stream):
            if delay_ms > 0:
                from . import _hip
                _hip.check(_hip.lib().fgnn_spin(int(delay_ms * 1e5), _hip.stream_ptr()))
            ops.stamp('step begin')
            fn()
            ops.stamp('step end')
        self.stamps, ops.STAMPS = ops.STAMPS, None
        if dot:
            self.graph.debug_dump(dot)

    def replay(self):
        self.graph.replay()
        if self.static_params:
            return
        from .mpnn import pointwise
        pointwise.note_state_change()       # the replayed kernels may have changed parameters / BatchNorm buffers
        pointwise.invalidate_casts()        # ... and with them the low-precision weight copies eager code reads next

    def stamp_report(self, file):
        """The stamps of the LAST replay, sorted by device time (microseconds from 'step begin')."""
        if not self.stamps:
            return
        torch.cuda.synchronize()
        v = self.stamps['buf'][:len(self.stamps['tags'])].cpu().tolist()
        t0 = v[0]
        for t, tag in sorted(zip(v, self.stamps['tags'])):
            print('stamp %10.2f us  %s' % ((t - t0) / 100.0, tag), file=file)

    __call__ = replay
