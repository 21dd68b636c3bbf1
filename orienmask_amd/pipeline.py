"""Several batches in flight: forward + postprocess of consecutive batches on alternating HIP streams.

Every convolution launch is a persistent grid that drains a tile queue; while its last tiles finish, part of the chip
idles (6-26 % of a launch, profiles/r02_experiments.md section 1), and nothing of the SAME batch can run there because
layer n+1 needs all of layer n.  A second batch has no such dependency: with two batches enqueued on two streams the
hardware places the other batch's workgroups on the compute units a draining launch frees.  Per-kernel efficiency is
unchanged (whole batches, unlike model.set_streams' sub-batches); each batch in flight owns a forward workspace
(model.workspace_slot) and a postprocess workspace.  Measured on MI355X at 32 x 544 x 544, forward only
(tools/pipelined_steps.py): fp32 29.6 -> 27.4 ms per batch, fp16 9.5 -> 8.4 ms.  End to end (bench.py) a third batch in flight
changes nothing in fp32 and adds 2.5 % in fp16, whose steps are a third as long; a fourth costs 3 %.

The reference runs one batch at a time (/root/reference/infer.py:153-156, /root/reference/trainer/tester.py:38-44: one
`model(image)` + `postprocess(predict)` per loader iteration, synchronising on `.cpu()` each time); its loop

    for image in loader:  dets = postprocess(model(image))

becomes

    pipe = InFlightPipeline(model, postprocess, depth=2)
    for dets in pipe.map(loader):  ...

with bit-identical detections in the same order (tests/test_hip_parity.py::test_in_flight_pipeline_matches_eager).
"""
import collections
import copy

import torch

from . import lib as _lib


class InFlightPipeline:
    def __init__(self, model, postprocess, depth=2, fuse_step=None):
        if int(depth) < 1:
            raise ValueError("depth must be >= 1")
        self.model = model.eval()
        self.depth = int(depth)
        # True: eval.launch_step -- the forward's C call launches decode + select on a second stream as soon as the box heads are
        # written (same bits); False: model(x), then postprocess.launch, on the slot's stream
        # Default (None): off.  Batches in flight on several streams already overlap one batch's select kernel with another's
        # convolutions, every fused batch drives TWO streams (the runtime maps streams onto four hardware queues) and pays two
        # event operations per step on the host: measured eagerly, same box, 32 x 544^2 two in flight 2134 / 2098 (off) against
        # 2110 / 2123 (on), three in flight 2133 against 2065, four single images in flight 569-737 against 486-496
        # (tools/inflight_ab.py).  The form pays where ONE stream replays a captured graph: graph.GraphedPipeline.
        if fuse_step is None:
            fuse_step = False
        self.fuse_step = bool(fuse_step) and hasattr(postprocess, "launch_step")
        # PRIVATE resources per batch in flight: forward workspace slots 1..depth (slot 0 stays the eager model(x) path's, which
        # runs on the caller's stream: sharing it would race with a pending batch on a side stream) and one copy of the
        # postprocess per slot -- same configuration, own workspace cache
        self._posts = []
        for _ in range(self.depth):
            p = copy.copy(postprocess)
            p._ws = {}
            self._posts.append(p)
        self._streams = {}
        self._pending = collections.deque()
        self._next = 0

    def __len__(self):
        return len(self._pending)

    def submit(self, image):
        """Enqueue forward + postprocess of one batch; returns at once (no host synchronisation).  At most `depth`
        batches may be pending; results come back from result() in submission order."""
        _lib.require_cuda_tensor(image, "image", torch.float32)
        if len(self._pending) >= self.depth:
            raise RuntimeError("InFlightPipeline: %d batches are already in flight; call result() first" % self.depth)
        dev = image.device
        slot = self._next
        self._next = (slot + 1) % self.depth
        streams = self._streams.setdefault(dev, [])
        while len(streams) < self.depth:
            streams.append(torch.cuda.Stream(device=dev))
        s = streams[slot]
        s.wait_stream(torch.cuda.current_stream(dev))          # the image was produced on the caller's stream
        with torch.cuda.stream(s), torch.no_grad(), self.model.workspace_slot(slot + 1):
            outs = (self._posts[slot].launch_step(self.model, image) if self.fuse_step
                    else self._posts[slot].launch(self.model(image)))
            done = torch.cuda.Event()
            done.record(s)
        image.record_stream(s)
        self._pending.append((slot, outs, done, dev))

    def result(self):
        """Detections of the oldest pending batch (list of dicts as OrienMaskYOLOPostProcess.apply returns them).  If that
        batch's forward reported OM_STATUS_SPLIT_RANGE, collect() re-runs it with fp32 operands on the caller's stream, in the
        batch's own workspace slot (it is free: the batch has finished)."""
        if not self._pending:
            raise RuntimeError("InFlightPipeline: nothing in flight")
        slot, outs, done, dev = self._pending.popleft()
        cur = torch.cuda.current_stream(dev)
        cur.wait_event(done)
        for t in outs[:5]:
            t.record_stream(cur)                               # the caller goes on using them on its own stream
        with torch.cuda.device(dev):
            return self._posts[slot].collect(outs)

    def map(self, batches):
        """Generator over the detections of every batch of `batches`, in order, keeping `depth` batches in flight."""
        for image in batches:
            if len(self._pending) == self.depth:
                yield self.result()
            self.submit(image)
        while self._pending:
            yield self.result()
