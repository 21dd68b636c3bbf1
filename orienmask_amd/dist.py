"""Multi-GPU plumbing: one process per GPU, batches sharded across ranks, weights broadcast once.

The reference has no multi-GPU inference (assert n_gpu <= 1, /root/reference/test.py:23,
/root/reference/infer.py:69); images are independent through forward and postprocess
(/root/reference/eval/orienmask_yolo_postprocess.py:75 loops per image), so the path shards with no
data-path collective.  The only collective is ONE broadcast of rank 0's weights over RCCL/xGMI at start-up (not in the
timed region): the raw fp32 state_dict -- 63,662,063 parameters + the BatchNorm running statistics, 254.9 MB, SURVEY.md 8e --
and every rank packs it ON ITS OWN DEVICE into the library's layouts (orienmask_amd/pack.py: BatchNorm folded in float64, the
Winograd transforms, the hi/lo fp16 pairs; under a second per rank).  Rounds 1-5 packed on rank 0 and broadcast the packed
blobs (1.16 GB + 0.66 GB in the default precision): 1.5 GB more xGMI traffic and rank 0's host-side pack for nothing -- the
device packer is bit-identical to the host's (tests/test_hip_parity.py::test_pack_on_device_equals_pack_on_host) and to itself
across ranks (checked by `verify=True`: one all-gather of a 64-bit checksum per blob).  Results are merged
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


def _float_entries(net):
    """The state_dict entries that travel, in state_dict order: every floating-point tensor (weights, biases, BatchNorm affine and
    running statistics).  `num_batches_tracked` (int64 counters, irrelevant in eval mode: /root/reference/model/base.py:113-128)
    stays local."""
    return [(k, v) for k, v in net.state_dict().items() if torch.is_tensor(v) and v.is_floating_point()]


def blob_checksum(blob):
    """64-bit checksum of a packed blob's bytes (sum of its 4-byte words as int64, and of the words weighted by position mod 251):
    equal blobs have equal checksums on every rank; used to verify that the per-rank packs agree."""
    w = blob.contiguous().view(torch.int16 if blob.element_size() == 2 else torch.int32).to(torch.int64)
    pos = (torch.arange(w.numel(), device=w.device, dtype=torch.int64) % 251) + 1
    return [int(w.sum().item()), int((w * pos).sum().item())]


def broadcast_packed_weights(net, device, src=0, stats=None, verify=False):
    """Broadcast rank `src`'s weights (it must hold the real state_dict) as ONE flat fp32 blob of the raw state_dict, load them
    into every rank's module, pack them on every rank's own device and bind the packed blobs of the net's precision.  `stats`
    (a dict) receives the bytes moved per rank, the number of blobs (1) and the wall time of the broadcast itself and of the
    per-rank pack (device-synchronised) -- what bench.py prints as `weight_broadcast`.  verify=True all-gathers a checksum of
    every packed blob and raises if the ranks disagree (stats['blobs_identical_across_ranks'])."""
    import time
    import torch.distributed as dist
    from . import lib as _lib
    from . import pack as _pack
    rank, world = world_info()
    h = net._ensure_handle()
    device = torch.device(device)
    on_gpu = device.type == "cuda"
    entries = _float_entries(net)
    numel = sum(v.numel() for _, v in entries)
    flat = None
    if rank == src:
        flat = torch.cat([v.detach().reshape(-1).to(device=device, dtype=torch.float32) for _, v in entries])
    if on_gpu:
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    flat = broadcast_blob(flat, numel, device, src)
    if on_gpu:
        torch.cuda.synchronize(device)
    bc_ms = (time.perf_counter() - t0) * 1e3
    # every rank (the source too: all ranks then run the SAME packer on the SAME bytes) takes its parameters from the blob ...
    sd = dict(net.state_dict())
    off = 0
    for k, v in entries:
        sd[k] = flat[off:off + v.numel()].view(v.shape)
        off += v.numel()
    # ... as device tensors: pack.py works on the device its input lives on
    t0 = time.perf_counter()
    blobs = {}
    L = _lib.load()
    blobs["f32"] = _pack.pack_state_dict(sd, net._layers, L.om_model_weight_floats(h))
    prec = getattr(net, "precision", "f32")
    if prec == "f16":       # the fp16 configuration needs the fp16 weight rows as well
        blobs["f16"] = _pack.pack_state_dict_f16(sd, net._layers, L.om_model_weight_halfs(h))
    if prec == "f32_split":     # split-operand mode: the hi/lo fp16 pairs of every layer's weights
        blobs["split"] = _pack.pack_state_dict_split(sd, net._layers, L.om_model_weight_split_words(h))
    if on_gpu:
        torch.cuda.synchronize(device)
    pack_ms = (time.perf_counter() - t0) * 1e3
    if rank != src:         # the module itself holds the weights too (a later precision switch packs from its state_dict)
        with torch.no_grad():
            own = net.state_dict()
            for k, _ in entries:
                own[k].copy_(sd[k].to(own[k].device))
    if on_gpu:
        net.bind_packed(blobs["f32"])
        if "f16" in blobs:
            net.bind_packed_f16(blobs["f16"])
        if "split" in blobs:
            net.bind_packed_split(blobs["split"])
    identical = None
    if verify:
        mine = {k: blob_checksum(b) for k, b in blobs.items()}
        if world > 1 and dist.is_available() and dist.is_initialized():
            every = [None] * world
            dist.all_gather_object(every, mine)
        else:
            every = [mine]
        identical = all(e == every[0] for e in every)
        if not identical:
            raise RuntimeError("the per-rank packs of the broadcast weights differ: %r" % (every,))
    if stats is not None:
        stats.update(bytes=numel * 4, ms=round(bc_ms, 3), blobs=1, pack_ms_per_rank=round(pack_ms, 3),
                     packed_bytes={k: b.numel() * b.element_size() for k, b in blobs.items()})
        if identical is not None:
            stats["blobs_identical_across_ranks"] = identical
    return blobs


def gather_detections(local, src_indices=None):
    """Host-side merge of per-rank result lists (any picklable objects) in rank order."""
    import torch.distributed as dist
    rank, world = world_info()
    if world == 1:
        return list(local)
    out = [None] * world
    dist.all_gather_object(out, list(local))
    return [d for part in out for d in part]
