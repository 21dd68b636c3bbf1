"""The reference's evaluation / inference loops on the HIP path.

``Tester.test`` mirrors /root/reference/trainer/tester.py:26-62 (what is timed, under which names, and how
fps is derived: 1000 * batch_size / mean batch ms); ``infer_loop`` mirrors the timed part of
/root/reference/infer.py:124-188 (10 warm-up iterations, 'Load data' / 'Forward & Postprocess' timers).
COCO metric computation (pycocotools) is outside the path; ``on_batch`` receives each batch's detections so
a caller can plug it in.  Batches come from any iterable yielding ``(image[B,3,H,W], ..., batch_info)``
tuples, e.g. ``SyntheticLoader``.
"""
import torch

from . import timer as _timer
from .dist import shard_range, world_info


class SyntheticLoader:
    """Deterministic stand-in for the COCO loader: n_images 544x544 images in batches (no dataset offline)."""

    def __init__(self, n_images, batch_size, size=(544, 544), seed=0, device="cuda"):
        from . import synth
        self.batch_size = batch_size
        rank, world = world_info()
        start, stop = shard_range(n_images, rank, world)            # each rank evaluates its own slice
        self._batches = []
        import torch
        for i, s in enumerate(range(start, stop, batch_size)):
            n = min(batch_size, stop - s)
            # one seed per IMAGE id: the images do not depend on the world size or on where a rank's batches begin, so the
            # shards of a sharded run are exactly the unsharded set (tests/test_dist_cpu.py compares every image)
            img = torch.cat([synth.synth_image_batch(seed + s + k, 1, size[0], size[1]) for k in range(n)], 0)
            info = [dict(id=s + k, height=size[0], width=size[1]) for k in range(n)]
            self._batches.append((img, None, info))
        self.device = device

    def __len__(self):
        return len(self._batches)

    def __iter__(self):
        return iter(self._batches)


class Tester:
    def __init__(self, model, postprocess, test_loader, device, on_batch=None):
        self.model = model
        self.postprocess = postprocess
        self.test_loader = test_loader
        self.device = device
        self.on_batch = on_batch

    def test(self, verbose=True, in_flight=1):
        """in_flight = 1: the reference's loop and timers.  in_flight > 1: the same batches through
        pipeline.InFlightPipeline (forward + postprocess of consecutive batches overlap on alternating HIP streams, so the two
        stages cannot be timed apart): one 'Forward & Postprocess' figure, wall time of the loop over its images; detections
        reach `on_batch` in the same order."""
        if in_flight > 1:
            return self._test_in_flight(verbose, in_flight)
        _timer.reset()
        _timer.cpu() if str(self.device) == "cpu" else _timer.cuda()
        self.model.eval()
        n_det = 0
        with torch.no_grad():
            for sample in self.test_loader:
                image = sample[0].to(self.device)
                batch_info = sample[2]
                with _timer.timer("Network Forward"):
                    predict = self.model(image)
                with _timer.timer("Postprocess"):
                    detections = self.postprocess(predict)
                n_det += sum(int(d["bbox"].shape[0]) for d in detections)
                if self.on_batch is not None:
                    with _timer.timer("Convert Format"):
                        self.on_batch(batch_info, detections)
        log = _timer.get_all_elapsed_time()
        bs = self.test_loader.batch_size
        stats = {k: dict(ms_per_image=v / bs, fps=1000.0 * bs / v) for k, v in log.items()}
        if verbose:
            print("Speed Statistics (batch size = {})".format(bs))
            for k, v in log.items():
                print("%s: %.3fms (%.3ffps)" % (k, v / bs, 1000 * bs / v))
        stats["detections"] = n_det
        return stats


    def test_and_gather(self, verbose=True, in_flight=1):
        """The multi-GPU form of `test`: every rank evaluates ITS slice of the loader (SyntheticLoader and
        tools/eval_val2017.py's Val2017Loader take contiguous slices by dist.shard_range) and whatever `on_batch` RETURNS per
        batch (a record or a list of records, any picklable objects) is merged in rank order on every rank -- rank-order
        concatenation of contiguous slices is dataset order.  This is /root/reference/trainer/trainer.py:175-181,201-205 (each
        rank writes _temp_coco_eval_<rank>.json, rank 0 concatenates them) without the files.  No collective in the step: one
        all_gather_object of host lists at the end.  Returns (this rank's stats, the merged records)."""
        from .dist import gather_detections
        records = []
        user = self.on_batch

        def collect(batch_info, detections):
            r = user(batch_info, detections) if user is not None else None
            if r is not None:
                records.extend(r if isinstance(r, list) else [r])

        self.on_batch = collect
        try:
            stats = self.test(verbose=verbose, in_flight=in_flight)
        finally:
            self.on_batch = user
        return stats, gather_detections(records)

    def _test_in_flight(self, verbose, depth):
        import time
        from .pipeline import InFlightPipeline
        self.model.eval()
        pipe = InFlightPipeline(self.model, self.postprocess, depth=depth)
        infos, n_det, n_img = [], 0, 0
        convert_ms = 0.0

        def batches():
            for sample in self.test_loader:
                infos.append(sample[2])
                yield sample[0].to(self.device)

        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        for i, detections in enumerate(pipe.map(batches())):
            n_det += sum(int(d["bbox"].shape[0]) for d in detections)
            n_img += len(detections)
            if self.on_batch is not None:
                c0 = time.perf_counter()
                self.on_batch(infos[i], detections)
                convert_ms += (time.perf_counter() - c0) * 1e3
        torch.cuda.synchronize(self.device)
        total_ms = (time.perf_counter() - t0) * 1e3 - convert_ms
        stats = {"Forward & Postprocess": dict(ms_per_image=total_ms / max(n_img, 1), fps=1000.0 * n_img / total_ms)}
        if self.on_batch is not None:
            stats["Convert Format"] = dict(ms_per_image=convert_ms / max(n_img, 1), fps=1000.0 * n_img / max(convert_ms, 1e-9))
        if verbose:
            print("Speed Statistics (batch size = {}, {} batches in flight)".format(self.test_loader.batch_size, depth))
            for k, v in stats.items():
                print("%s: %.3fms (%.3ffps)" % (k, v["ms_per_image"], v["fps"]))
        stats["detections"] = n_det
        return stats


def infer_loop(model, transform, postprocess, images, device, warmup=10, size_divisor=32, use_graph=True, latency_mode=True):
    """images: list of [h,w,3] float32 tensors (what cv2.imread + cvtColor give infer.py).
    Returns (list of per-image detections, list of pad_info, timer log in ms).

    use_graph (default): forward + postprocess of every network input SHAPE that occurs run as one captured hipGraph
    (graph.GraphedPipeline: ~95 kernel launches become one graph launch; bit-identical detections, tests/test_hip_parity.py::
    test_tester_and_infer_loops); a postprocess built for one image size only sees that size (infer.py resizes to 544 x 544
    first), other shapes fall back to eager launches.  Detections of a graphed call are views of graph-owned buffers that the
    next call overwrites, so they are cloned here (the reference returns fresh tensors).

    latency_mode (default): this is the reference's ONE-IMAGE loop (infer.py:143-172), so the model's latency mode is on while
    it runs (model.set_latency_mode: direct 3x3 convolutions instead of the fused Winograd kernel for a batch this small) and
    restored afterwards."""
    _timer.reset()
    _timer.cuda()
    model.eval()
    results, pads = [], []
    graphs = {}
    restore = None
    if latency_mode and getattr(model, "precision", None) == "f32_split" and hasattr(model, "set_latency_mode"):
        restore = model.latency_cells
        model.set_latency_mode(True)
    # Only this package's postprocess with its own NMS can be captured: capture needs the kernel-only `launch` / `collect` split
    # and its workspace, and a foreign nms_func runs on the host in the middle of the step (a synchronisation, which stream
    # capture refuses).  Any other callable -- the reference's contract is just `postprocess(model(x))` -- runs eagerly.
    graphable = (use_graph and all(hasattr(postprocess, a) for a in ("launch", "collect", "_ws"))
                 and getattr(postprocess, "nms_thresh", None) is not None)

    def run(x):
        if not graphable:
            return postprocess(model(x))
        key = tuple(x.shape)
        if key not in graphs:
            from .graph import GraphedPipeline
            g = None
            if len(graphs) < 4:      # a handful of shapes at most
                try:
                    g = GraphedPipeline(model, postprocess, x)
                except RuntimeError as e:      # a capture that failed leaves nothing behind: this shape runs eagerly
                    import warnings
                    warnings.warn("infer_loop: hipGraph capture failed for input shape %s (%s); running it eagerly" % (key, e))
            graphs[key] = g
        g = graphs[key]
        if g is None:
            return postprocess(model(x))
        return [{k: v.clone() for k, v in d.items()} for d in g(x)]

    try:
        with torch.no_grad():
            if warmup and images:
                x, _ = transform.padded(images[0].to(device).unsqueeze(0), size_divisor)
                for _ in range(warmup):
                    run(x)
            with _timer.timer("Main Loop"):
                for img in images:
                    with _timer.timer("Load data"):
                        x, pad_info = transform.padded(img.to(device).unsqueeze(0), size_divisor)
                    with _timer.timer("Forward & Postprocess"):
                        det = run(x)
                    results.append(det[0])
                    pads.append(pad_info)
    finally:
        if restore is not None:
            model.set_latency_mode(restore > 0, restore or None)
    return results, pads, _timer.get_all_elapsed_time()
