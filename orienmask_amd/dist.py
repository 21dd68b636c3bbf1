"""Multi-GPU plumbing: one process per GPU, batches sharded across ranks, weights broadcast once.

The reference has no multi-GPU inference (assert n_gpu <= 1, /root/reference/test.py:23,
/root/reference/infer.py:69); images are independent through forward and postprocess
(/root/reference/eval/orienmask_yolo_postprocess.py:75 loops per image), so the path shards with no
data-path collective.  The only collective is one broadcast of rank 0's packed weight blob
(63.67 M weights plus the Winograd transforms of the 3x3 layers: 1.16 GB; in the default split-operand precision also
the hi/lo fp16 form of every layer's weights, 0.66 GB: 1.82 GB in two blobs) over RCCL/xGMI at start-up -- not in the timed region.  Results are merged
on the host per rank, as the reference's validation does with its _temp_coco_eval_%d.json files
(/root/reference/trainer/trainer.py:175-181,201-205).
"""
import torch


def world_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_items, rank, world):
    """Contiguous slice [start, stop) of n_items owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def broadcast_blob(blob, numel, device, src=0, dtype=torch.float32):
    """Broadcast a flat blob (float32, or float16 for the fp16 weights) from `src`; other ranks pass blob=None."""
    import torch.distributed as dist
    rank, world = world_info()
    if rank == src:
        if blob is None or blob.numel() != numel or blob.dtype != dtype:
            raise ValueError("source rank must provide the %d-element %s blob" % (numel, dtype))
        blob = blob.to(device).contiguous()
    else:
        blob = torch.empty(numel, dtype=dtype, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.broadcast(blob, src=src)
    return blob


def broadcast_packed_weights(net, device, src=0, stats=None):
    """Pack on `src` (which must hold the real state_dict), broadcast, bind on every rank.  `stats` (a dict) receives the
    bytes moved per rank, the number of blobs and the wall time of the broadcasts themselves (device-synchronised; packing on
    the source's host is not in it) -- what bench.py prints as `weight_broadcast`."""
    import time
    from . import lib as _lib
    from . import pack as _pack
    rank, _ = world_info()
    h = net._ensure_handle()
    acc = dict(bytes=0, ms=0.0, blobs=0)

    def timed_broadcast(blob, numel, dtype=torch.float32):
        if torch.device(device).type == "cuda":
            torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        out = broadcast_blob(blob, numel, device, src, dtype=dtype)
        if torch.device(device).type == "cuda":
            torch.cuda.synchronize(device)
        acc["ms"] += (time.perf_counter() - t0) * 1e3
        acc["bytes"] += out.numel() * out.element_size()
        acc["blobs"] += 1
        return out

    numel = _lib.load().om_model_weight_floats(h)
    blob = _pack.pack_state_dict(net.state_dict(), net._layers, numel) if rank == src else None
    blob = timed_broadcast(blob, numel)
    net.bind_packed(blob)
    if getattr(net, "precision", "f32") == "f16":       # the fp16 configuration needs the fp16 weight rows as well
        n16 = _lib.load().om_model_weight_halfs(h)
        b16 = _pack.pack_state_dict_f16(net.state_dict(), net._layers, n16) if rank == src else None
        net.bind_packed_f16(timed_broadcast(b16, n16, dtype=torch.float16))
    if getattr(net, "precision", "f32") == "f32_split":     # split-operand mode: the hi/lo fp16 pairs of the F(2x4) weights
        ns = _lib.load().om_model_weight_split_words(h)
        bs = _pack.pack_state_dict_split(net.state_dict(), net._layers, ns) if rank == src else None
        net.bind_packed_split(timed_broadcast(bs, ns))
    if stats is not None:
        stats.update(bytes=acc["bytes"], ms=round(acc["ms"], 3), blobs=acc["blobs"])
    return blob


def gather_detections(local, src_indices=None):
    """Host-side merge of per-rank result lists (any picklable objects) in rank order."""
    import torch.distributed as dist
    rank, world = world_info()
    if world == 1:
        return list(local)
    out = [None] * world
    dist.all_gather_object(out, list(local))
    return [d for part in out for d in part]
