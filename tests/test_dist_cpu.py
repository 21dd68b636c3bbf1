"""CPU, world_size 2 over gloo: the sharding and weight-broadcast plumbing of the N>1 path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from orienmask_amd.dist import shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from orienmask_amd import dist as omd, lib as omlib, pack, synth
    from orienmask_amd.model import OrienMaskYOLOFPNPlus
    net = OrienMaskYOLOFPNPlus(3, 80).eval()                     # the plugin's default precision: split operands
    h = net._ensure_handle()
    L = omlib.load()
    if rank == 0:                               # only rank 0 has the real weights
        net.load_state_dict(synth.synth_state_dict(9), strict=True)
    # ONE broadcast of the raw fp32 state_dict (~255 MB), every rank packs for itself; verify=True all-gathers a
    # checksum of every packed blob and raises if the ranks disagree
    stats = {}
    blobs = omd.broadcast_packed_weights(net, torch.device("cpu"), src=0, stats=stats, verify=True)
    ref_sd = synth.synth_state_dict(9)
    ok_blob = stats["blobs"] == 1 and 2.54e8 < stats["bytes"] < 2.56e8 and stats["blobs_identical_across_ranks"] is True
    ok_blob = ok_blob and bool(torch.equal(blobs["f32"], pack.pack_state_dict(ref_sd, net._layers, L.om_model_weight_floats(h))))
    ns = L.om_model_weight_split_words(h)
    ok_blob = ok_blob and ns > 0 and bool(torch.equal(blobs["split"].view(torch.int32),
                                                      pack.pack_state_dict_split(ref_sd, net._layers, ns).view(torch.int32)))
    # every rank's MODULE holds rank 0's weights afterwards (a later precision switch packs from its own state_dict)
    own = net.state_dict()
    ok_blob = ok_blob and all(torch.equal(own[k], v) for k, v in ref_sd.items() if torch.is_tensor(v) and v.is_floating_point())
    # the fp16 weight rows of the fp16-activation configuration: packed per rank from the same broadcast
    net.set_precision("f16")
    blobs16 = omd.broadcast_packed_weights(net, torch.device("cpu"), src=0, verify=True)
    ok_blob = ok_blob and bool(torch.equal(blobs16["f16"], pack.pack_state_dict_f16(ref_sd, net._layers, L.om_model_weight_halfs(h))))
    # the building block still moves any blob (fp16 too)
    probe = torch.arange(1000, dtype=torch.float16) if rank == 0 else None
    ok_blob = ok_blob and bool(torch.equal(omd.broadcast_blob(probe, 1000, torch.device("cpu"), src=0, dtype=torch.float16),
                                           torch.arange(1000, dtype=torch.float16)))
    start, stop = omd.shard_range(67, rank, world)
    merged = omd.gather_detections([{"image": i} for i in range(start, stop)])
    q.put((rank, ok_blob, start, stop, [m["image"] for m in merged]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 32, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_broadcast_and_merge(built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in results] == [True, True]              # both ranks hold rank 0's weights and packed them identically
    assert (results[0][2], results[0][3], results[1][2], results[1][3]) == (0, 34, 34, 67)
    assert results[0][4] == list(range(67)) and results[1][4] == list(range(67))


def _tester_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from orienmask_amd.tester import SyntheticLoader, Tester

    class Model:            # the loop's contract only: eval(), __call__ (the HIP model needs a GPU; the sharding logic does not)
        def eval(self):
            return self

        def __call__(self, x):
            return x

    def post(pred):         # one "detection" per image carrying a fingerprint of that image's pixels
        return [dict(bbox=pred[b].sum().reshape(1, 1)) for b in range(pred.shape[0])]

    loader = SyntheticLoader(11, 4, size=(32, 32), seed=3, device="cpu")       # 11 images: ranks get 6 and 5, last batches ragged
    tester = Tester(Model(), post, loader, "cpu",
                    on_batch=lambda infos, dets: [dict(id=i["id"], v=float(d["bbox"][0, 0])) for i, d in zip(infos, dets)])
    stats, merged = tester.test_and_gather(verbose=False)
    q.put((rank, stats["detections"], merged))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_tester_merges_in_dataset_order(built):
    """Tester.test_and_gather: every rank evaluates its slice, the per-batch records come back merged in
    dataset order on every rank -- /root/reference/trainer/trainer.py:175-181,201-205 without the per-rank json files."""
    from orienmask_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tester_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in results] == [6, 5]
    assert results[0][2] == results[1][2] and [m["id"] for m in results[0][2]] == list(range(11))
    # the single-process loader sees the same images (SyntheticLoader seeds every image by its id): ALL merged fingerprints are
    # the unsharded ones, whatever the ranks' batch boundaries
    want = [float(synth.synth_image_batch(3 + i, 1, 32, 32)[0].sum()) for i in range(11)]
    got = [m["v"] for m in results[0][2]]
    assert len(got) == 11 and all(abs(g - w) <= 1e-6 * abs(w) for g, w in zip(got, want)), (got, want)
    assert len(set(round(w, 3) for w in want)) == 11        # and the fingerprints do tell the images apart


# ---- `python bench.py --gpus N` means N: launcher resolution, and the self-spawned job on gloo
def _bench_module():
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    spec = importlib.util.spec_from_file_location("om_bench_under_test", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, path


def test_bench_gpus_flag_is_binding():
    import pytest
    bench, path = _bench_module()
    assert bench.resolve_launch(1, {}, []) == ("run", 1)
    assert bench.resolve_launch(8, {"WORLD_SIZE": "8"}, []) == ("run", 8)
    action, cmd = bench.resolve_launch(8, {}, ["--gpus", "8", "--steps", "3"], visible_gpus=8)
    assert action == "spawn"
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and "127.0.0.1" in cmd and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    assert os.path.samefile(cmd[cmd.index("--master-port") + 2], path)
    for gpus, env, visible in ((2, {"WORLD_SIZE": "1"}, None), (1, {"WORLD_SIZE": "2"}, None), (8, {"WORLD_SIZE": "4"}, None),
                               (0, {}, None), (2, {}, 1), (8, {}, 0)):
        with pytest.raises(SystemExit) as e:
            bench.resolve_launch(gpus, env, [], visible_gpus=visible)
        assert e.value.code not in (0, None) and "--gpus" in str(e.value.code)


def test_bench_gpus_2_without_a_launcher_runs_two_ranks(built):
    """The command the driver types for N = 1 must work verbatim for N > 1: no launcher, `--gpus 2` -> two ranks, a line with
    n_gpus 2 (here on gloo with --plumbing-only: no GPU in this container), the slowest rank's time in it."""
    import json
    import subprocess
    import sys
    _, path = _bench_module()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, path, "--gpus", "2", "--plumbing-only"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                 # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["broadcast"]["intact"] and d["broadcast"]["bytes"] == 4 << 20
    assert d["rank_ms"]["max"] >= 19.0 > d["rank_ms"]["min"] >= 9.0  # rank 1 slept 20 ms, rank 0 10 ms
    # a launcher that disagrees with --gpus is refused by every rank
    r = subprocess.run([sys.executable, path, "--gpus", "2", "--plumbing-only"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    # and without GPUs the real bench fails loudly instead of printing a 1-GPU line
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, path, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "n_gpus" not in r.stdout
