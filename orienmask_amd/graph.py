"""hipGraph capture of the hot path (forward + postprocess kernels) for a fixed input shape.

Capture goes through torch's stream capture (`torch.cuda.CUDAGraph`), which records every kernel the C ABI
launches on the capturing stream.  The library is capture-safe: no allocation, no synchronisation, no memcpy and
no memset on the hot path -- counters are cleared by a kernel (`launch_zero_words`), because a captured
`hipMemsetAsync` node left part of its range uncleared when the graph was replayed (ROCm 7.2; DESIGN.md "hipGraph").

    pipe = GraphedPipeline(model, postprocess, example_input)     # warms up, then captures
    detections = pipe(image)                                      # copy-in, one graph launch, one D2H of the counts

The tensors in `detections` are views of graph-owned buffers: the next call overwrites them (clone to keep).

The captured kernels hold RAW device pointers into the model's workspace, the postprocess workspace and the packed
weight blobs.  The pipeline keeps its own references to those tensors, so an eager `model(x)` / `postprocess(pred)` at
another shape (which drops them from the model's caches) cannot hand their memory to someone else; and a weight change
after capture (`load_state_dict`, `.to()`, `set_precision`) makes the next call raise instead of replaying stale weights.

Measured on MI355X (tools/graph_bench.py) replay and eager take the same time at every batch size
(B=1 5.25 ms, B=4 7.55 ms, B=32 34.2 ms): the ~130 launches are issued far ahead of the GPU, so the step is
GPU-bound even at B=1 and the graph only removes host work (useful when the host thread is busy with decoding).
The reference has no equivalent (it runs eager torch ops); results are bit-identical to the eager call sequence
`postprocess(model(image))` (tests/test_hip_parity.py::test_graphed_pipeline_matches_eager).
"""
import torch


class GraphedPipeline:
    def __init__(self, model, postprocess, example_input, warmup=3, fuse_step=True):
        if not example_input.is_cuda:
            raise RuntimeError("GraphedPipeline needs a CUDA example input (no CPU fallback)")
        self.model = model.eval()
        self.post = postprocess
        self.fuse_step = fuse_step        # False: model(x) then postprocess.launch on one stream (A/B of eval.launch_step)
        self.static_in = example_input.detach().clone().contiguous()
        dev = self.static_in.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                       # allocates workspaces and packed weights outside the capture
                self._launch()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self._launch()
        # everything the captured launches point into (see the module docstring)
        self._precision = model.precision
        self._weights = (model._packed, model._packed16 if model.precision == "f16" else None,
                         model._packed_split if model.precision == "f32_split" else None)
        self._keepalive = list(model._workspace.values()) + list(postprocess._ws.values()) + [w for w in self._weights if w is not None]

    def _check_bindings(self):
        m = self.model
        now = (m._packed, m._packed16 if m.precision == "f16" else None,
               m._packed_split if m.precision == "f32_split" else None)
        if m.precision != self._precision or any(a is not b for a, b in zip(now, self._weights)):
            raise RuntimeError("GraphedPipeline: the model's weights or precision changed after capture; build a new "
                               "GraphedPipeline (the captured kernels read the packed blobs that were bound at capture time)")

    def _launch(self):
        # kernels only, no host synchronisation; decode + select beside the orientation branch (eval.launch_step)
        step = getattr(self.post, "launch_step", None) if self.fuse_step else None
        self._outs = step(self.model, self.static_in) if step is not None else self.post.launch(self.model(self.static_in))

    def __call__(self, image):
        if image.shape != self.static_in.shape:
            raise ValueError("captured for shape %s, got %s" % (tuple(self.static_in.shape), tuple(image.shape)))
        self._check_bindings()
        self.static_in.copy_(image, non_blocking=True)
        self.graph.replay()
        return self.post.collect(self._outs)
