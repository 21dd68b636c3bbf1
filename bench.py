"""Headline benchmark: images/sec end-to-end (forward + postprocess) at 544x544, bs=32 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is what the reference times around `model(image); postprocess(predict)`
(/root/reference/trainer/tester.py:39-44, /root/reference/infer.py:154-156) on one batch of 32
synthetic 544x544 images already resident in HBM (BASELINE.json configs[2]).  Each rank runs its own
batch (weak scaling, no collective in the timed region); rank 0's packed weights are broadcast once
over RCCL before timing.  Rank 0 prints ONE JSON line.

roofline:     the dominant kernel (default precision: the fused F(4,3) 3x3 convolution on the fp16 matrix cores with split
              fp32 operands).  achieved = ALGORITHMIC flops -- the direct convolution's 2 * B * Ho * Wo * cout * cin * k^2 of the
              layers the kernel runs (SURVEY.md 8d's per-image count x the images of a launch) -- / their summed duration
              INCLUDING a layer's input-transform pre-pass where there is one, measured live with HIP events on the launch
              stream over the timed steps (om_profile_*); peak = the dense peak of the matrix instruction the kernel uses
              (2.5 PFLOP/s fp16, 157.3 TFLOP/s f32-input: MI355X_MICROARCH.md); frac = achieved / peak.  What the matrix pipe
              EXECUTES for those flops (x3 for split operands, x1/2 for F(4,3) along the rows, x1/3 for F(2x4)) is reported
              next to it as achieved_executed / executed_frac.  The HBM view of the same launches is `roofline.hbm`, the
              forward's `forward_hbm_*` / `conv_stack_hbm_pmc`.
              traffic = HBM bytes per launch from the committed rocprofv3 PMC passes of this same bench -- printed only
              when that file was measured with the library binary that is running now (sha256), else null.
cpu_baseline: the CPU oracle (torch-CPU restatement of the reference, oracle/) on the host cores: thread-count sweep,
              forward and postprocess timed separately at bs=1 and on the bench batch, N=1 only.  Checker code, timed
              beside the product, never part of it.
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402

ANCHORS_YOLOV4 = [[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]]
ANCHOR_MASK = [[6, 7, 8], [3, 4, 5], [0, 1, 2]]
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_F16_MFMA_TFLOPS = 2500.0      # dense fp16/bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
PROFILE_TAG = "r06"                # profiles/<tag>_pmc_traffic*.json: the PMC passes whose `traffic` this build may quote
WEIGHT_SEED, OBJ_BIAS, HEAD_GAIN = 3, -16.0, 4.0
OBJ_BIAS_SPARSE = -18.5     # a few tens of detections per image (-18: 67, -19: 11, -20: 3, <= -24: none)


def post_config(h, w):
    return dict(grid_size=[[h // 32, w // 32], [h // 16, w // 16], [h // 8, w // 8]], image_size=[h, w],
                anchors=ANCHORS_YOLOV4, anchor_mask=ANCHOR_MASK, num_classes=80, conf_thresh=0.005,
                nms_pre=400, nms_post=100, orien_thresh=0.3)


def cpu_baseline(sd, x_cpu, f16=False, budget_s=30.0):
    """Oracle forward and postprocess on the host cores: thread sweep {8, 16, 32, 64} on one image, then with
    the best count: bs=1 forward / postprocess, and the bench batch itself (or as many images of it as the budget allows)."""
    from oracle import orienmask_ref as R
    fwd = R.forward_f16 if f16 else R.forward
    h, w = x_cpu.shape[2], x_cpu.shape[3]
    pc = post_config(h, w)
    post = R.PostProcessOracle(pc["grid_size"], pc["image_size"], pc["anchors"], pc["anchor_mask"], 80,
                               conf_thresh=pc["conf_thresh"])
    ncpu = os.cpu_count() or 1
    t_start = time.perf_counter()
    one = x_cpu[:1]
    sweep = {}
    # {8, 16, 32, 64} and "all" only up to 64: on the 256-thread GPU host one 544x544 forward took 69.7 s with all threads
    # (oversubscribed MKL-DNN), against 0.18-0.27 s with 8-32 -- measured once (profiles/r02_bench.json), not repeated every run
    for nt in sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu), min(64, ncpu)}):
        torch.set_num_threads(nt)
        fwd(sd, one)                                                   # warm-up at this thread count
        t0 = time.perf_counter()
        pred = fwd(sd, one)
        sweep[nt] = time.perf_counter() - t0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    t0 = time.perf_counter(); pred = fwd(sd, one); f1 = time.perf_counter() - t0
    t0 = time.perf_counter(); post(pred); p1 = time.perf_counter() - t0
    left = budget_s - (time.perf_counter() - t_start)
    nb = int(max(2, min(x_cpu.shape[0], left / max(f1 * 0.8 + p1, 1e-3))))      # batched forward is a little cheaper per image
    batch = x_cpu[:nb]
    t0 = time.perf_counter(); pred = fwd(sd, batch); fb = time.perf_counter() - t0
    t0 = time.perf_counter(); post(pred); pb = time.perf_counter() - t0
    return dict(value=round(nb / (fb + pb), 4), unit="images/s", cores=best, kind="port",
                sample="torch-CPU oracle (oracle/orienmask_ref.py), forward + postprocess of %d of the bench batch's %d images "
                       "(544x544) in one call, %d threads (best of the sweep)" % (nb, x_cpu.shape[0], best),
                host_cpus=ncpu,
                thread_sweep_forward_bs1_ms={str(k): round(v * 1e3, 1) for k, v in sweep.items()},
                bs1=dict(forward_ms=round(f1 * 1e3, 1), postprocess_ms=round(p1 * 1e3, 1),
                         images_per_s=round(1.0 / (f1 + p1), 4)),
                batch=dict(images=nb, forward_ms=round(fb * 1e3, 1), postprocess_ms=round(pb * 1e3, 1),
                           images_per_s=round(nb / (fb + pb), 4)))


class ClockSampler:
    """The shader clock of THIS GPU while a timed region runs, from the driver's own table (sysfs pp_dpm_sclk: the level marked
    `*` is the current clock), sampled every few milliseconds by a host thread.  Under dense fp16 matrix work on non-zero data
    MI355X does not hold its 2.4 GHz: the peaks of /opt/skills/guides/MI355X_MICROARCH.md are quoted at 2.4 GHz, so the line
    also says what the matrix pipe could do at the clock it actually ran at (profiles/r05_experiments.md section 4)."""

    def __init__(self, dev):
        import glob, threading
        self.path = None
        try:
            pr = torch.cuda.get_device_properties(dev)
            bus = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for card in glob.glob("/sys/class/drm/card*/device"):
                if os.path.basename(os.path.realpath(card)) == bus and os.path.exists(os.path.join(card, "pp_dpm_sclk")):
                    self.path = os.path.join(card, "pp_dpm_sclk")
        except Exception:
            self.path = None
        self._threading = threading
        self.samples, self._stop, self._thread = [], None, None

    def _read(self):
        try:
            for line in open(self.path).read().splitlines():
                if line.rstrip().endswith("*"):
                    return float(line.split(":")[1].strip().split("M")[0])
        except Exception:
            return None
        return None

    def start(self):
        self.samples = []
        if self.path is None:
            return self
        self._stop = self._threading.Event()

        def run():
            while not self._stop.is_set():
                v = self._read()
                if v is not None:
                    self.samples.append(v)
                self._stop.wait(0.004)

        self._thread = self._threading.Thread(target=run, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None
        v = sorted(self.samples)
        if not v:
            return None
        return dict(mean=round(sum(v) / len(v), 1), min=v[0], median=v[len(v) // 2], max=v[-1], samples=len(v))


def lib_sha256():
    from orienmask_amd import lib as omlib
    return hashlib.sha256(open(omlib.LIB_PATH, "rb").read()).hexdigest()


def post_occupancy(B, post_cfg_):
    """Occupancy of the three postprocess kernels against the gfx950 limits (8 waves per SIMD = 32 per CU, 160 KiB of LDS):
    what the registers and LDS allow per CU, and what the launch geometry actually puts on the chip."""
    from orienmask_amd import lib as omlib
    L = omlib.load()
    names = ["post_decode_kernel", "post_select_kernel", "post_mask_kernel"]
    h, w = post_cfg_["image_size"]
    npairs = sum(3 * gh * gw for gh, gw in post_cfg_["grid_size"]) * 80
    grids = [((npairs + 2047) // 2048) * B, B, ((h // 4 + 1) * (w // 16) + 255) // 256 * B * 9]
    out = {}
    for i, name in enumerate(names):
        t, v, l, nb = (ctypes.c_int(0) for _ in range(4))
        omlib.check(L.om_post_kernel_occupancy(i, ctypes.byref(t), ctypes.byref(v), ctypes.byref(l), ctypes.byref(nb)),
                    "om_post_kernel_occupancy")
        waves_per_wg = t.value // 64
        alloc = (v.value + 7) // 8 * 8                                  # register allocation granule (MI355X_MICROARCH.md)
        by_regs = min(8, 512 // max(alloc, 1))                          # waves per SIMD the register file allows
        by_lds = ((160 * 1024) // l.value) * waves_per_wg / 4.0 if l.value else 99.0
        limits = {"wave slots": 8.0, "registers": float(by_regs), "lds": by_lds}
        wg_per_cu = nb.value
        waves_per_simd = min(8.0, wg_per_cu * waves_per_wg / 4.0)
        limited_by = min(limits, key=limits.get)
        resident_wgs = min(grids[i], wg_per_cu * 256)
        out[name] = dict(threads_per_workgroup=t.value, vgprs=v.value, lds_bytes_per_workgroup=l.value,
                         workgroups_per_cu=wg_per_cu, waves_per_simd=waves_per_simd, limit_waves_per_simd=8,
                         occupancy_frac=round(waves_per_simd / 8.0, 3),
                         limited_by=limited_by,
                         grid_workgroups=grids[i],
                         chip_wave_slots_used_frac=round(resident_wgs * waves_per_wg / (256 * 32.0), 4))
    return out


def measure_neighbours(dev, dets, B):
    """Side measurements of the two rows next to the path; not part of `value`."""
    from orienmask_amd import synth
    from orienmask_amd.coco_format import COCOFormatter
    from orienmask_amd.transform import FastCOCOTransform
    out = {}
    tf = FastCOCOTransform([FastCOCOTransform.Resize((544, 544)), FastCOCOTransform.Normalize((0, 0, 0), (255, 255, 255))])
    img = synth.synth_photo_batch(5, B, 480, 640).to(dev)
    for _ in range(3):
        tf.padded(img)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        x, _ = tf.padded(img)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byts = img.numel() * 4 + x.numel() * 4
    out["preprocess"] = dict(kernel="preprocess_kernel", workload="%d x 480x640x3 float HWC -> [%d,3,544,544]" % (B, B),
                             ms=round(ms, 4), images_per_s=round(B / ms * 1e3, 1), bound="hbm", unit="GB/s",
                             achieved=round(byts / ms / 1e6, 1), peak=PEAK_HBM_GBS, frac=round(byts / ms / 1e6 / PEAK_HBM_GBS, 4))
    fmt = COCOFormatter(list(range(1, 81)), with_mask=True)
    infos = [dict(id=i, height=480, width=640, collate_pad=[0, 0, 0, 0, 544, 544]) for i in range(len(dets))]
    fmt.to_coco_format(infos[:2], dets[:2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = min(8, len(dets))
    res = fmt.to_coco_format(infos[:n], dets[:n])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["coco_format"] = dict(kernels="recover_bbox_kernel + recover_rle_kernel", workload="%d images x %d masks 544x544 -> 480x640 RLE"
                              % (n, len(res["segm"]) // max(n, 1)), ms_per_image=round(dt / n * 1e3, 3),
                              note="wall time incl. the host-side string packing; the reference copies 29.6 MB of masks per image to the host instead")
    return out


def resolve_launch(gpus, env, argv, visible_gpus=None):
    """What `python bench.py --gpus N` means.  Returns ("run", world) when this process is one rank
    of a job whose size agrees with --gpus, or ("spawn", cmd) when no launcher set WORLD_SIZE and N > 1: the command that
    re-executes this script as N ranks under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1).  A launcher
    whose WORLD_SIZE differs from --gpus, N < 1, or N above the visible GPUs is an error -- never a silent N = 1 run."""
    if gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1 (got %d)" % gpus)
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; pass --gpus %d or launch %d ranks"
                             % (gpus, world, world, gpus))
        return "run", world
    if gpus == 1:
        return "run", 1
    if visible_gpus is not None and visible_gpus < gpus:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible on this node" % (gpus, visible_gpus))
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return "spawn", cmd


def plumbing_only(args, world, rank):
    """`--plumbing-only`: the launcher / rendezvous / broadcast / max-over-ranks timing of the N > 1 path on gloo with CPU tensors
    -- what a box without N GPUs can check of `bench.py --gpus N` (tests/test_dist_cpu.py).  No kernel runs; the line says so."""
    import torch.distributed as dist
    from orienmask_amd.dist import broadcast_blob
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1 << 20
    blob = torch.arange(n, dtype=torch.float32) if rank == 0 else None
    t0 = time.perf_counter()
    got = broadcast_blob(blob, n, torch.device("cpu"), src=0)
    bc_ms = (time.perf_counter() - t0) * 1e3
    ok = bool(got[12345].item() == 12345.0)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))                                   # ranks take different times: the line must carry the max
    el = time.perf_counter() - t0
    per_rank = [el]
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64)
        lst = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(lst, t)
        per_rank = [v.item() for v in lst]
    if rank == 0:
        print(json.dumps(dict(metric="plumbing only (no kernel ran): launcher, rendezvous, broadcast and rank reduction of "
                                     "bench.py --gpus N on gloo", value=None, unit=None, n_gpus=world, rccl_ranks=world,
                              broadcast=dict(bytes=n * 4, ms=round(bc_ms, 3), intact=ok),
                              rank_ms=dict(min=round(min(per_rank) * 1e3, 3), max=round(max(per_rank) * 1e3, 3)),
                              data="none")))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-solo-reference", action="store_true",
                    help="N > 1: skip rank 0's solo run of the same steps (the denominator of `scaling_efficiency`)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="no GPU work: run only the N-rank launcher / rendezvous / broadcast / timing reduction on gloo (CPU test of --gpus N)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (metric is quoted at 32)")
    ap.add_argument("--size", type=int, default=544)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-compare", action="store_true", help="split mode: skip the extra timed region with fp32 operands")
    ap.add_argument("--no-f16-compare", action="store_true",
                    help="split mode: skip the extra timed region in the fp16-activation configuration (BASELINE configs[4])")
    ap.add_argument("--no-extras", action="store_true", help="skip the preprocess / COCO-format side measurements")
    ap.add_argument("--latency-mode", action="store_true",
                    help="f32_split: model.set_latency_mode(True) for the whole run (direct 3x3 convolutions below ~4 images per batch)")
    ap.add_argument("--fuse-step", action="store_true",
                    help="A/B: the in-flight pipeline behind `value` goes through eval.launch_step (decode + select on a second stream "
                         "beside the orientation branch; same bits) instead of model(x) + postprocess.launch on one stream per batch; "
                         "the graph replays of `small_batches` always do")
    ap.add_argument("--latency-ksplit", type=int, default=None,
                    help="with --latency-mode: most parts a small launch's k loop is cut into (om_model_set_latency_ksplit; default: the library's)")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the bs = 1 / bs = 8 latency figures (`small_batches`)")
    ap.add_argument("--layers", action="store_true", help="also print the per-layer table to stderr")
    ap.add_argument("--heads", choices=("dense", "sparse", "allpass"), default="dense",
                    help="head statistics of the seeded weights (SURVEY.md 8d Config 3 asks for both): dense = every image yields "
                         ">400 candidates and 100 detections (the default); sparse = a few tens of detections; allpass = SURVEY 8c's "
                         "dense WORST case, random-init-like heads (objectness bias 0, logits ~0): every one of the 1 456 560 "
                         "(candidate, class) pairs passes conf_thresh, so the decode evaluates all of them and the select kernel takes "
                         "its three-pass radix path over 5.8 MB of keys per image")
    ap.add_argument("--obj-bias", type=float, default=None, help="override the objectness bias of the synthetic weights")
    ap.add_argument("--streams", type=int, default=1,
                    help="forward as N sub-batches on N HIP streams in the timed region (model.set_streams); the default 1 is "
                         "what the roofline figures assume (per-kernel events time overlapping kernels otherwise)")
    ap.add_argument("--in-flight", type=int, default=None,
                    help="batches in flight in the timed region behind `value` (orienmask_amd.pipeline.InFlightPipeline: whole "
                         "batches on alternating HIP streams, so that one batch's draining tile queues overlap the other's "
                         "kernels).  1 = the reference's loop, one batch at a time; that region is ALWAYS timed too (it carries "
                         "the per-kernel events of the roofline object) and reported as `one_batch_in_flight`.  Default: 2 in fp32, "
                         "3 in fp16 (its steps are a third as long and the host's read of the counts weighs more: 3790 against 3700 "
                         "images/s; in fp32 a third batch changes nothing, a fourth costs 3 %)")
    ap.add_argument("--replicated-concat", action="store_true",
                    help="A/B: f32_split with the routes / skips stored replicated into the concat buffers (set_upsample_on_read(False)) "
                         "instead of read up-sampled by the 1x1 layer behind the concat")
    ap.add_argument("--lib", default=None,
                    help="path of another build of liborienmask_hip.so to load instead of the in-tree one (A/B runs of two "
                         "builds on the same GPU box: tools/ab_bench.sh)")
    ap.add_argument("--dtype", choices=("f32", "f32_split", "f16"), default=None,
                    help="default: the plugin's own default precision (orienmask_amd.model.DEFAULT_PRECISION = f32_split), i.e. what "
                         "build(config['model'], orienmask_amd.model) runs.  f32_split: fp32 tensors and fp32 accumulation, every convolution product computed from hi/lo "
                         "fp16 pairs of its fp32 operands on the fp16 matrix pipe (three MFMAs per product group; error against "
                         "float64 equal to the fp32-operand kernels: profiles/r02_split_error.json) -- the headline metric, with "
                         "the fp32-operand time reported beside it as `f32_operands`.  f32: fp32 operands on "
                         "v_mfma_f32_32x32x2_f32 only.  f16: BASELINE configs[4], fp16 activations and weights with fp32 "
                         "accumulation -- a separate, clearly labelled line")
    args = ap.parse_args()
    from orienmask_amd.model import DEFAULT_PRECISION
    if args.dtype is None:
        args.dtype = DEFAULT_PRECISION
    if args.in_flight is None:
        args.in_flight = 3 if args.dtype == "f16" else 2

    # --gpus N means N: a launcher's WORLD_SIZE must agree with it; without a launcher and N > 1 this process re-executes
    # itself as N ranks (one per GPU) under torch.distributed.run and returns their exit code
    action, what = resolve_launch(args.gpus, os.environ, sys.argv[1:],
                                  None if args.plumbing_only else (torch.cuda.device_count() if torch.cuda.is_available() else 0))
    if action == "spawn":
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool: RCCL needs it
        env.setdefault("OMP_NUM_THREADS", "8")
        raise SystemExit(subprocess.call(what, env=env))
    world = what
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.plumbing_only:
        return plumbing_only(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("OM_BENCH_FORCE_DIST") == "1"     # the env switch exercises the RCCL calls on one GPU
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.lib:
        from orienmask_amd import lib as _omlib
        _omlib.LIB_PATH = os.path.abspath(args.lib)
    from orienmask_amd import arch, synth
    from orienmask_amd.dist import broadcast_packed_weights
    from orienmask_amd.eval import OrienMaskYOLOPostProcess
    from orienmask_amd.model import OrienMaskYOLOFPNPlus

    H = W = args.size
    B = args.batch
    obj_bias = args.obj_bias if args.obj_bias is not None else {"dense": OBJ_BIAS, "sparse": OBJ_BIAS_SPARSE, "allpass": 0.0}[args.heads]
    head_gain = 0.25 if args.heads == "allpass" else HEAD_GAIN
    net = OrienMaskYOLOFPNPlus(3, 80).eval().set_precision(args.dtype)
    f16 = args.dtype == "f16"
    split = args.dtype == "f32_split"
    sd = None
    if rank == 0:
        sd = synth.synth_state_dict(WEIGHT_SEED, obj_bias=obj_bias, head_gain=head_gain)
        net.load_state_dict(sd, strict=True)
    bc_stats = {}
    # ONE RCCL broadcast of rank 0's raw fp32 state_dict (~255 MB), packed on every rank's own device; untimed
    broadcast_packed_weights(net, dev, src=0, stats=bc_stats, verify=True)
    if args.replicated_concat:
        net.set_upsample_on_read(False)
    if args.latency_mode:
        net.set_latency_mode(True, ksplit=args.latency_ksplit)
    post = OrienMaskYOLOPostProcess(device=dev, **post_config(H, W))
    x_cpu = synth.synth_image_batch(1000 + rank, B, H, W)
    x = x_cpu.to(dev)
    # two input batches taken in turn: a step never re-reads the tensor the step before it left in the 256 MiB Infinity Cache
    xs = [x, synth.synth_image_batch(2000 + rank, B, H, W).to(dev)]
    step_no = [0]

    def step():
        step_no[0] += 1
        with torch.no_grad():
            return post(net(xs[step_no[0] & 1]))

    def batches(n):
        import itertools
        return itertools.islice(itertools.cycle(xs), n)

    def over_ranks(seconds):
        """(max, [per-rank seconds]) of a timed region: the job's time is its slowest rank's"""
        if not use_dist:
            return seconds, [seconds]
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        lst = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(lst, t)
        per = [v.item() for v in lst]
        return max(per), per

    for _ in range(args.warmup):
        dets = step()
    torch.cuda.synchronize()

    # ---- N > 1: rank 0 alone (every other GPU idle at the barrier) runs the K steps one batch at a time and with batches in
    # flight: the SAME binary's single-GPU figures, which `scaling_efficiency` divides by
    solo = None
    if use_dist and (world > 1 or os.environ.get("OM_BENCH_FORCE_DIST") == "1") and not args.no_solo_reference:
        dist.barrier()
        if rank == 0:
            import itertools
            from orienmask_amd.pipeline import InFlightPipeline
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                dets = step()
            torch.cuda.synchronize()
            solo_serial = time.perf_counter() - t0
            solo_flight = solo_serial
            if args.in_flight > 1:
                pipe0 = InFlightPipeline(net, post, depth=args.in_flight, fuse_step=args.fuse_step)
                for dets in pipe0.map(batches(2 * args.in_flight)):
                    pass
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for dets in pipe0.map(batches(args.steps)):
                    pass
                torch.cuda.synchronize()
                solo_flight = time.perf_counter() - t0
                del pipe0
            solo = dict(value=round(B * args.steps / solo_flight, 2), one_batch_in_flight=round(B * args.steps / solo_serial, 2))
        dist.barrier()

    # ---- untimed pass with events around EVERY layer: the per-kernel table, and which kernel dominates
    specs = {s.name: s for s in arch.fpnplus_convs()}
    kernel_of = dict(net.layer_kernels(B, H, W))
    n_prof = 3
    net.profile_enable(True)
    post_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_prof)]
    for i in range(n_prof):
        with torch.no_grad():
            pred = net(x)
            post_ev[i][0].record()
            outs = post.launch(pred)
            post_ev[i][1].record()                                 # kernels only: collect()'s host read is not GPU time
            dets = post.collect(outs)
    torch.cuda.synchronize()
    layer_ms, n_fw = net.profile_read()
    post_ms = sum(a.elapsed_time(b) for a, b in post_ev) / n_prof
    by_kernel = {}
    for name, ms, pre in layer_ms:
        by_kernel[kernel_of[name]] = by_kernel.get(kernel_of[name], 0.0) + ms + pre
    dom = max(by_kernel, key=by_kernel.get)
    dom_layers = [name for name, _, _ in layer_ms if kernel_of[name] == dom]

    # ---- timed region: K steps, HIP events only around the dominant kernel's launches (an event pair costs a few
    # microseconds of GPU time; around all ~90 layers that was 3 % of the step)
    net.profile_enable_layers(dom_layers)
    net.set_streams(args.streams)
    for _ in range(2 if args.streams > 1 else 0):
        step()                                                     # allocate the per-stream workspaces outside the timing
    if args.streams > 1:
        net.profile_enable_layers(dom_layers)                      # drop the warm-up's events
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    clock = ClockSampler(dev)
    sclk_idle = clock._read() if clock.path else None
    clock.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        dets = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    sclk_serial = clock.stop()
    sclk_flight = None
    elapsed, per_rank_serial = over_ranks(elapsed)
    per_rank = per_rank_serial
    timed_ms, timed_fw = net.profile_read()
    net.profile_enable(False)
    net.set_streams(1)

    # ---- the same K steps with args.in_flight whole batches in flight (no per-kernel events: kernels of different batches
    # overlap here, their individual durations say nothing)
    elapsed_serial = elapsed
    if args.in_flight > 1:
        import itertools
        from orienmask_amd.pipeline import InFlightPipeline
        pipe = InFlightPipeline(net, post, depth=args.in_flight, fuse_step=args.fuse_step)
        for dets in pipe.map(batches(2 * args.in_flight)):                 # allocates the per-slot workspaces, untimed
            pass
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        clock.start()
        t0 = time.perf_counter()
        for dets in pipe.map(batches(args.steps)):
            pass
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        sclk_flight = clock.stop()
        elapsed, per_rank = over_ranks(elapsed)
    # ---- split mode: the same K steps with fp32 operands (v_mfma_f32_32x32x2_f32 everywhere), for comparison
    f32_operands = None
    if split and not args.no_f32_compare:
        net.set_precision("f32")
        def timed(fn):
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            return over_ranks(time.perf_counter() - t0_)[0]
        for _ in range(2):
            step()
        e1 = timed(lambda: [step() for _ in range(args.steps)])
        e2 = e1
        if args.in_flight > 1:
            for _ in pipe.map(batches(2 * args.in_flight)):
                pass
            e2 = timed(lambda: [None for _ in pipe.map(batches(args.steps))])
        f32_operands = dict(value=round(world * B * args.steps / e2, 2), ms_per_step=round(e2 / args.steps * 1e3, 3),
                            one_batch_in_flight=round(world * B * args.steps / e1, 2),
                            note="the same K steps with precision 'f32': fp32 operands on v_mfma_f32_32x32x2_f32 (157 TFLOP/s "
                                 "peak) in every convolution; same tensors in HBM, same postprocess")
        net.set_precision(args.dtype)
    # ---- split mode: the same K steps in the fp16-activation configuration (BASELINE configs[4]: fp16 activations and weights,
    # fp32 accumulate -- NARROWER arithmetic than the headline, reported beside it so that the driver's run times it too)
    f16_config = None
    if split and not args.no_f16_compare:
        import itertools
        from orienmask_amd.pipeline import InFlightPipeline
        net.set_precision("f16")
        broadcast_packed_weights(net, dev, src=0)                  # the fp16 weight rows (untimed)
        pipe16 = InFlightPipeline(net, post, depth=3)
        for _ in range(2):
            step()
        for _ in pipe16.map(batches(6)):
            pass
        def timed16(fn):
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            return over_ranks(time.perf_counter() - t0_)[0]
        clock.start()
        e1 = timed16(lambda: [step() for _ in range(args.steps)])
        sclk16_serial = clock.stop()
        clock.start()
        e3 = timed16(lambda: [None for _ in pipe16.map(batches(args.steps))])
        sclk16_flight = clock.stop()
        # this configuration's own roofline: HIP events around every layer of three forwards (untimed), the kernel with the most
        # time, its executed flops (direct convolution on v_mfma_f32_32x32x16_f16: executed = algorithmic) against the dense
        # fp16 peak, and the forward's algorithmic bytes (2-byte activations and weights) against HBM
        net.profile_enable(True)
        for _ in range(3):
            with torch.no_grad():
                net(xs[0])
        torch.cuda.synchronize()
        lms16, nfw16 = net.profile_read()
        net.profile_enable(False)
        kof16 = dict(net.layer_kernels(B, H, W))
        kt = {}
        for name, ms, pre in lms16:
            wk = arch.layer_work(specs[name], B, H, W)
            t = kt.setdefault(kof16[name], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            t["ms"] += (ms + pre) / nfw16; t["flops"] += wk["flops"]; t["bytes"] += wk["bytes"] / 2; t["launches"] += 1
        dom16 = max(kt, key=lambda k: kt[k]["ms"])
        fwd16_ms = sum(t["ms"] for t in kt.values())
        d16 = kt[dom16]
        traffic16 = None
        try:
            pmc16 = json.load(open(os.path.join(REPO, "profiles", PROFILE_TAG + "_pmc_traffic_f16.json")))
            if pmc16.get("_meta", {}).get("lib_sha256") == lib_sha256():
                keys = [k for k in pmc16 if k.replace(" ", "").startswith(dom16.replace(" ", "").rstrip(">")) and
                        "hbm_bytes_per_launch_corrected" in pmc16[k]]
                if keys:
                    nl = sum(pmc16[k]["launches"] for k in keys)
                    traffic16 = round(sum(pmc16[k]["hbm_bytes_per_launch_corrected"] * pmc16[k]["launches"] for k in keys) / nl)
        except Exception:
            pass
        f16_roofline = dict(bound="mfma", kernel=dom16, launches_per_step=d16["launches"], kernel_ms_per_step=round(d16["ms"], 3),
                            achieved=round(d16["flops"] / (d16["ms"] * 1e-3) / 1e12, 2), peak=PEAK_F16_MFMA_TFLOPS, unit="TFLOP/s",
                            frac=round(d16["flops"] / (d16["ms"] * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
                            algorithmic_bytes_per_launch=round(d16["bytes"] / d16["launches"]), traffic=traffic16,
                            forward_kernels_ms_per_step=round(fwd16_ms, 3),
                            forward_tflops=round(sum(t["flops"] for t in kt.values()) / (fwd16_ms * 1e-3) / 1e12, 2),
                            forward_hbm_algorithmic_gbs=round(sum(t["bytes"] for t in kt.values()) / (fwd16_ms * 1e-3) / 1e9, 1),
                            forward_hbm_frac=round(sum(t["bytes"] for t in kt.values()) / (fwd16_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                            step_hbm_algorithmic_frac=round(sum(t["bytes"] for t in kt.values()) / (e3 / args.steps) / 1e9 / PEAK_HBM_GBS, 4),
                            measured="HIP events around every layer of three untimed one-batch-at-a-time forwards in this run; "
                                     "traffic = PMC bytes per launch when profiles/%s_pmc_traffic_f16.json was measured with this "
                                     "library binary, else null" % PROFILE_TAG)
        if sclk16_serial:
            f16_roofline["sclk_mhz"] = dict(one_batch_in_flight=sclk16_serial, batches_in_flight=sclk16_flight, nominal=2400)
        f16_config = dict(value=round(world * B * args.steps / e3, 2), ms_per_step=round(e3 / args.steps * 1e3, 3),
                          batches_in_flight=3, one_batch_in_flight=round(world * B * args.steps / e1, 2), roofline=f16_roofline,
                          note="BASELINE configs[4] at this batch size: the same K steps with precision 'f16' (fp16 activations and "
                               "convolution weights, fp32 accumulate, fp32 heads and postprocess; `bench.py --dtype f16` prints its own "
                               "roofline).  Narrower arithmetic than the headline and unpinned against the reference (DESIGN.md 3.3): "
                               "never `value`")
        # ... and at the batch size BASELINE configs[4] names (64 images per GPU): a few steps with three batches in flight
        if B == 32 and args.batch == 32:
            try:
                xs64 = [torch.cat([xs[0], xs[1]]), torch.cat([xs[1], xs[0]])]
                n64 = max(6, args.steps // 3)
                for _ in pipe16.map(itertools.islice(itertools.cycle(xs64), 6)):
                    pass
                e64 = timed16(lambda: [None for _ in pipe16.map(itertools.islice(itertools.cycle(xs64), n64))])
                f16_config["bs64"] = dict(value=round(world * 64 * n64 / e64, 2), ms_per_step=round(e64 / n64 * 1e3, 3), steps=n64,
                                          batches_in_flight=3, note="64 images per GPU per step, the configuration's own batch size")
                del xs64
            except RuntimeError as err:      # (out of memory on a small device: the figure is an extra)
                f16_config["bs64"] = dict(error=str(err)[:200])
        del pipe16
        net.set_precision(args.dtype)
    # ---- small batches (the reference's own published metric is bs = 1 FPS: README.md:5, infer.py:143-172): one image / eight
    # images per step, one at a time (the reference's loop) through the hipGraph of forward + postprocess that infer_loop uses,
    # eagerly, and with four batches in flight; same weights, same precision
    small = None
    if not args.no_small_batch and args.streams == 1:
        import itertools
        from orienmask_amd.graph import GraphedPipeline
        from orienmask_amd.pipeline import InFlightPipeline
        small = {}
        for bsz in (1, 8):
            xsb = [t[:bsz].contiguous() for t in xs]
            n_it = max(20, args.steps)
            def time_loop(fn, n):
                for _ in range(3):
                    fn(0)
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for i in range(n):
                    fn(i)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0_) / n
            with torch.no_grad():
                eager_s = time_loop(lambda i: post(net(xsb[i & 1])), n_it)
                lat_s = None
                if split:           # the latency mode infer_loop switches on (model.set_latency_mode): direct 3x3 layers below ~4 images
                    net.set_latency_mode(True)
                    gpl = GraphedPipeline(net, post, xsb[0])
                    lat_s = time_loop(lambda i: gpl(xsb[i & 1]), n_it)
                    del gpl
                    net.set_latency_mode(False)
                gp = GraphedPipeline(net, post, xsb[0])
                graph_s = time_loop(lambda i: gp(xsb[i & 1]), n_it)
                del gp
                pipe4 = InFlightPipeline(net, post, depth=4)
                for _ in pipe4.map(itertools.islice(itertools.cycle(xsb), 8)):
                    pass
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for _ in pipe4.map(itertools.islice(itertools.cycle(xsb), 4 * n_it)):
                    pass
                torch.cuda.synchronize()
                fl_s = (time.perf_counter() - t0_) / (4 * n_it)
                del pipe4
            small["bs%d" % bsz] = dict(ms_per_step_one_at_a_time_graph=round(graph_s * 1e3, 3), images_per_s_graph=round(bsz / graph_s, 1),
                                       ms_per_step_one_at_a_time_eager=round(eager_s * 1e3, 3), images_per_s_eager=round(bsz / eager_s, 1),
                                       images_per_s_four_in_flight=round(bsz / fl_s, 1),
                                       ms_per_step_latency_mode_graph=round(lat_s * 1e3, 3) if lat_s else None,
                                       images_per_s_latency_mode=round(bsz / lat_s, 1) if lat_s else None)
        small["note"] = ("batches of 1 and 8 images per GPU with this run's weights and precision: one at a time through the captured "
                         "hipGraph (orienmask_amd.graph.GraphedPipeline, what infer_loop runs for a fixed shape), one at a time eagerly, "
                         "with four batches in flight (InFlightPipeline), and -- f32_split -- one at a time through the graph in the model's "
                         "latency mode (model.set_latency_mode, what tester.infer_loop switches on: stride-1 3x3 layers as direct "
                         "convolutions where the fused kernel would have <= 128 tiles, split-K parts in the implicit GEMM's small launches)")
    timed_fw //= max(args.streams, 1)                              # one om_forward per sub-batch
    dom_main_ms = sum(ms for name, ms, pre in timed_ms if name in set(dom_layers)) / timed_fw      # per step, all launches
    dom_timed_ms = dom_main_ms + sum(pre for name, ms, pre in timed_ms if name in set(dom_layers)) / timed_fw

    if rank == 0:
        kern = {}

        def acc(name, ms, flops, exec_flops, byts, pre_ms=0.0):
            t = kern.setdefault(name, dict(ms=0.0, pre_ms=0.0, flops=0.0, exec_flops=0.0, bytes=0.0, launches=0))
            t["ms"] += ms; t["pre_ms"] += pre_ms; t["flops"] += flops; t["exec_flops"] += exec_flops; t["bytes"] += byts
            t["launches"] += 1

        def reduction(k):
            # multiplies the matrix pipe executes per direct-convolution multiply: F(2x4,3x3) 24 per 8 outputs instead of 72,
            # F(2x2,3x3) 16 per 4 outputs instead of 36
            # split operands: three fp16 matrix instructions per product group (F(2x4): 3 x 1/3 = the direct-convolution count)
            # the fused F(4,3)-along-the-rows form: 18 products per 4 outputs instead of 36, times three fp16 instructions
            if k.startswith("wino14"):
                return 1.0 / 1.5
            if "split" in k:
                return 1.0 if k.startswith("wino24") else 1.0 / 3.0
            return 3.0 if k.startswith("wino24") else (2.25 if k.startswith("wino") else 1.0)

        pre_kernel = {}
        prev_k = None
        for name, ms, pre in layer_ms:
            wk = arch.layer_work(specs[name], B, H, W)
            k = kernel_of[name]
            if k.startswith("(in the previous"):
                # backbone.conv2.0 inside conv_stem2_split_kernel (launched for backbone.conv1): its work -- three fp16 matrix
                # instructions per product group -- and its algorithmic bytes are charged to that kernel, no launch of its own
                t = kern[prev_k]
                t["flops"] += wk["flops"]; t["exec_flops"] += 3.0 * wk["flops"]; t["bytes"] += wk["bytes"]; t["ms"] += ms / n_fw
                continue
            prev_k = k
            # fp16 configuration: every activation and weight element is 2 bytes (the fp32 heads are < 1 % of the bytes)
            red = 1.0 if k.startswith("conv_stem2") else reduction(k)      # conv1's own products run on the vector ALUs
            acc(k, ms / n_fw, wk["flops"], wk["flops"] / red, wk["bytes"] / (2 if f16 else 1), pre / n_fw)
            if pre > 0 and k.startswith("wino14_wide"):
                # the two-kernel wide form of the fused 3x3 kernel (round 6): its pre-pass reads the layer's input and writes the
                # transformed input, 1.5x the activation over the padded rows and whole tile columns
                hw = H // arch.layer_div(specs[name])
                pre_kernel[k] = "wino14_v_kernel"
                acc("wino14_v_kernel", pre / n_fw, 0.0, 0.0, 4.0 * B * specs[name].cin * (hw * hw + 1.5 * (hw + 2) * 4 * ((hw + 3) // 4)))
            if pre > 0 and (k.startswith("wino_gemm") or k.startswith("wino24_gemm")):
                vx = 3.0 if k.startswith("wino24") else 4.0       # transformed input: 3x / 4x the activation, plus reading it
                pk = "wino24_input_kernel" if k.startswith("wino24") else "wino_input_kernel"
                pre_kernel[k] = pk
                acc(pk, pre / n_fw, 0.0, 0.0, (1.0 + vx) * 4 * B * (H // arch.layer_div(specs[name])) ** 2 * specs[name].cin)
        d = kern[dom]
        fwd_ms = sum(t["ms"] for t in kern.values())              # pre-pass kernels are their own entries
        # the dominant kernel's rate comes from the events recorded INSIDE the timed region; a layer's input-transform
        # pre-pass belongs to the same convolution and is charged to it
        achieved_alg = d["flops"] / (dom_timed_ms * 1e-3) / 1e12
        executed = d["exec_flops"] / (dom_timed_ms * 1e-3) / 1e12
        executed_main_only = d["exec_flops"] / (dom_main_ms * 1e-3) / 1e12
        total_flops = sum(t["flops"] for t in kern.values())
        total_exec = sum(t["exec_flops"] for t in kern.values())
        total_bytes = sum(t["bytes"] for t in kern.values() if t["flops"] > 0)
        # ---- HBM traffic from the committed PMC passes, only if they were measured with THIS library binary
        traffic, traffic_src, conv_stack = None, None, None
        pmc_file = os.path.join(REPO, "profiles", PROFILE_TAG + ("_pmc_traffic_f16.json" if f16 else
                                ("_pmc_traffic_f32_split.json" if split else "_pmc_traffic.json")))
        try:
            pmc = json.load(open(pmc_file))
            meta = pmc.pop("_meta", {})
            norm = lambda k: k.replace(" ", "")
            def find(kname):
                """PMC record of a kernel as bench.py names it; a kernel compiled in several variants (e.g. the whole-tile and
                the stream-K form of one GEMM, <false> / <true> in the symbol) is the launch-weighted mean of its variants."""
                keys = [k for k in pmc if norm(k).startswith(norm(kname).rstrip(">"))] or \
                       [k for k in pmc if k.split("<")[0] == kname.split("<")[0]] or \
                       [k for k in pmc if k.startswith("_Z") and kname.split("<")[0] in k]      # a symbol rocprofv3 left mangled
                keys = [k for k in keys if "hbm_bytes_per_launch_corrected" in pmc[k]]
                if not keys:
                    return None
                n = sum(pmc[k]["launches"] for k in keys)
                return dict(launches=n, variants=keys,
                            hbm_bytes_per_launch_corrected=sum(pmc[k]["hbm_bytes_per_launch_corrected"] * pmc[k]["launches"] for k in keys) / n)
            if meta.get("lib_sha256") != lib_sha256():
                traffic_src = ("%s was measured with another build of liborienmask_hip.so (sha256 %s...): not reported; rerun "
                               "tools/pmc_traffic.sh" % (os.path.relpath(pmc_file, REPO), str(meta.get("lib_sha256"))[:12]))
            elif meta.get("batch") != B or meta.get("size") != H:
                traffic_src = "%s was measured at another problem size" % os.path.relpath(pmc_file, REPO)
            else:
                e = find(dom)
                if e:
                    traffic = round(e["hbm_bytes_per_launch_corrected"])
                    if dom in pre_kernel and find(pre_kernel[dom]):
                        traffic += round(find(pre_kernel[dom])["hbm_bytes_per_launch_corrected"])
                    traffic_src = ("%s: (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch%s, separate rocprofv3 --pmc passes of this "
                                   "bench with this library binary (FETCH_SIZE doubled per MI355X_MICROARCH.md)"
                                   % (os.path.relpath(pmc_file, REPO), " incl. the input-transform pre-pass" if dom in pre_kernel else ""))
                tot_b, tot_ms, missing = 0.0, 0.0, []
                for kname, t in kern.items():
                    e = find(kname)
                    if e is None:
                        missing.append(kname)
                        continue
                    tot_b += e["hbm_bytes_per_launch_corrected"] * t["launches"]
                    tot_ms += t["ms"]
                conv_stack = dict(gbs=round(tot_b / (tot_ms * 1e-3) / 1e9, 1), frac_of_8tbs=round(tot_b / (tot_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                  bytes_per_step=round(tot_b), kernels_ms_per_step=round(tot_ms, 3), kernels_without_counters=missing,
                                  note="rocprofv3 PMC HBM bytes of every convolution kernel of the forward / their summed durations in this run")
        except FileNotFoundError:
            traffic_src = "no %s" % os.path.relpath(pmc_file, REPO)
        except Exception as e:      # a malformed file must not take the bench line down
            traffic_src = "could not read %s: %s" % (os.path.relpath(pmc_file, REPO), e)
        peak_tf = PEAK_F16_MFMA_TFLOPS if (f16 or split) else PEAK_F32_MFMA_TFLOPS
        alg_bytes_per_launch = d["bytes"] / d["launches"]      # layer-fused model: input once, output once, residual once, weights
        # key order: the driver's record keeps the first keys of this object -- the figures a reader compares come first, prose last
        # `achieved` / `frac` count ALGORITHMIC flops (the direct convolution's 2 * B * Ho * Wo * cout * cin * 9 per launch:
        # SURVEY.md 8d's per-image figure x the images of a launch) -- what the matrix pipe EXECUTES for them (x3 for split
        # operands, x1/2 for F(4,3) along the rows) is `achieved_executed` / `executed_frac` (task 2)
        roofline = dict(bound="mfma", achieved=round(achieved_alg, 2), peak=peak_tf, unit="TFLOP/s",
                        frac=round(achieved_alg / peak_tf, 4), traffic=traffic,
                        frac_counts="ALGORITHMIC (direct-convolution) flops / time / peak; executed_frac counts what the matrix pipe "
                                    "executes for them (split operands x3, F(4,3) x1/2)" +
                                    ("" if (split or f16) else "; with fp32 operands the Winograd F(2x4) kernel executes 1/3 of the "
                                     "algorithmic flops, so `frac` may exceed 1 here -- executed_frac is this mode's fraction of the pipe"),
                        achieved_executed=round(executed, 2), executed_frac=round(executed / peak_tf, 4),
                        kernel=dom, launches_per_step=d["launches"], avg_launch_ms=round(dom_main_ms / d["launches"], 4),
                        kernel_ms_per_step=round(dom_timed_ms, 3),
                        algorithmic_bytes_per_launch=round(alg_bytes_per_launch),
                        traffic_over_algorithmic=round(traffic / alg_bytes_per_launch, 3) if traffic else None,
                        one_batch_in_flight_images_per_s=round(world * B * args.steps / elapsed_serial, 2),
                        one_batch_in_flight_ms_per_step=round(elapsed_serial / args.steps * 1e3, 3),
                        traffic_source=traffic_src,
                        pre_pass_kernel=pre_kernel.get(dom),
                        avg_launch_ms_with_pre_pass=round(dom_timed_ms / d["launches"], 4),
                        measured="HIP events on the launch stream around every launch of this kernel (and of its pre-pass) inside the "
                                 "timed one-batch-in-flight region; the other per-kernel figures come from an untimed pass with events around all layers",
                        achieved_without_pre_pass=round(executed_main_only, 2),
                        note=("fp16 operands, fp32 accumulate on v_mfma_f32_32x32x16_f16 (dense peak 2.5 PFLOP/s); direct "
                              "convolution, executed = algorithmic") if f16 else
                             ("achieved_executed = flops the fp16 matrix pipe executed (split operands: three v_mfma_f32_32x32x16_f16 per "
                              "product group; the fused kernel's F(4,3) along the rows multiplies 4.5 of the direct convolution's 9 "
                              "products per output, so executed = 3 x 1/2 of the direct-convolution flops; the input transform is "
                              "inside the kernel, there is no pre-pass) over the kernel's time, against the dense fp16 peak.  The "
                              "matrix pipe's minimum time for these launches (executed flops / 2.5 PFLOP/s) and HBM's (algorithmic "
                              "bytes / 8 TB/s) are both ~1/6 of the measured time: what binds is the CU's vector-memory request path "
                              "(DESIGN.md 3.6, profiles/r03_pmc_wino14_*.txt); `hbm` holds the memory-side view of the same launches")
                             if split and dom.startswith("wino14") else
                             ("achieved_executed = flops the fp16 matrix pipe executed (split operands: three v_mfma_f32_32x32x16_f16 per "
                              "product group, i.e. 3 x 1/3 of the direct-convolution flops for Winograd F(2x4,3x3)) over the time of "
                              "the GEMM kernel AND its input-transform pre-pass, against the dense fp16 peak") if split else
                             ("achieved_executed = flops the f32 matrix pipe executed (exact fp32 MFMA; Winograd F(2x4,3x3) runs 1/3 of the "
                              "direct-convolution multiplies) over the time of the GEMM kernel AND its input-transform pre-pass; "
                              "achieved counts direct-convolution flops over the same time"),
                        forward_kernels_ms_per_step=round(fwd_ms, 3), postprocess_ms_per_step=round(post_ms, 3),
                        forward_tflops_algorithmic=round(total_flops / (fwd_ms * 1e-3) / 1e12, 2),
                        forward_tflops_executed=round(total_exec / (fwd_ms * 1e-3) / 1e12, 2),
                        forward_executed_frac=round(total_exec / (fwd_ms * 1e-3) / 1e12 / peak_tf, 4),
                        forward_hbm_algorithmic_gbs=round(total_bytes / (fwd_ms * 1e-3) / 1e9, 1),
                        forward_hbm_frac=round(total_bytes / (fwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                        conv_stack_hbm_pmc=conv_stack,
                        postprocess_occupancy=post_occupancy(B, post_config(H, W)),
                        binding=("fp16: matrix pipe and HBM are within 2x of each other (SURVEY.md 8d); forward_hbm_frac is the "
                                 "north_star's HBM-roofline figure") if f16 else
                                ("split operands: 5.3x the fp32 matrix rate, so the forward's matrix-pipe minimum (executed flops / "
                                 "2.5 PFLOP/s) and its HBM minimum (algorithmic bytes / 8 TB/s) are of the same size; since round 3 the "
                                 "transformed input of the 3x3 layers stays on chip and the measured HBM traffic (conv_stack_hbm_pmc) is "
                                 "~1.3x the algorithmic bytes: the 3x3 kernel is bound by the CU's outstanding-request limit, the 1x1 "
                                 "layers at 136^2 / 68^2 by HBM") if split else
                                "fp32 FLOPs on the f32-input matrix cores; the HBM bound is several x further away")
        if sclk_serial:
            # data, not a roofline: the shader clock this GPU held during the timed regions (the peaks are quoted at 2400 MHz)
            roofline["sclk_mhz"] = dict(one_batch_in_flight=sclk_serial, batches_in_flight=sclk_flight, idle_before=sclk_idle, nominal=2400,
                                        source="sysfs pp_dpm_sclk of this GPU, sampled every 4 ms during the timed regions")
        front = ["bound", "achieved", "peak", "unit", "frac", "traffic", "frac_counts", "achieved_executed", "executed_frac", "kernel",
                 "launches_per_step", "avg_launch_ms", "kernel_ms_per_step", "algorithmic_bytes_per_launch", "traffic_over_algorithmic",
                 "one_batch_in_flight_images_per_s", "one_batch_in_flight_ms_per_step", "forward_kernels_ms_per_step",
                 "postprocess_ms_per_step", "forward_tflops_algorithmic", "forward_tflops_executed", "forward_executed_frac",
                 "forward_hbm_algorithmic_gbs", "forward_hbm_frac", "sclk_mhz"]
        front = [k for k in front if k in roofline]
        roofline = {**{k: roofline[k] for k in front}, **{k: v for k, v in roofline.items() if k not in front}}
        if split:
            roofline["hbm"] = dict(bound="hbm", unit="GB/s", peak=PEAK_HBM_GBS,
                                   achieved=round(d["bytes"] / (dom_timed_ms * 1e-3) / 1e9, 1),
                                   frac=round(d["bytes"] / (dom_timed_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                   traffic_gbs=round(traffic * d["launches"] / (dom_timed_ms * 1e-3) / 1e9, 1) if traffic else None,
                                   traffic_frac=round(traffic * d["launches"] / (dom_timed_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if traffic else None,
                                   note="the same launches against HBM: achieved = algorithmic bytes per launch (layer input, output, "
                                        "residual and weights once) / time; traffic_* = the PMC bytes (`traffic`) / time (round 2's "
                                        "two-kernel form moved the transformed input through HBM: 4.26x the algorithmic bytes)")
        if args.layers:
            for name, ms, pre in layer_ms:
                wk = arch.layer_work(specs[name], B, H, W)
                tot = (ms + pre) / n_fw
                print("%-28s %8.3f ms (pre %6.3f)  %7.2f TF  %7.1f GB/s  %s" % (
                    name, tot, pre / n_fw, wk["flops"] / (tot * 1e-3) / 1e12, wk["bytes"] / (tot * 1e-3) / 1e9,
                    kernel_of[name]), file=sys.stderr)
            for k, t in sorted(kern.items()):
                print("%-34s %3d launches %8.3f ms %7.2f TF algorithmic %7.2f TF executed %7.1f GB/s" % (
                    k, t["launches"], t["ms"], t["flops"] / (t["ms"] * 1e-3) / 1e12,
                    t["exec_flops"] / (t["ms"] * 1e-3) / 1e12, t["bytes"] / (t["ms"] * 1e-3) / 1e9), file=sys.stderr)
            print("forward kernels %.3f ms, postprocess %.3f ms" % (fwd_ms, post_ms), file=sys.stderr)
        total_images = world * B * args.steps
        metric = "images/sec end-to-end (544^2, bs=32) forward+postprocess"
        if f16:
            metric += " [fp16 activations, fp32 accumulate: BASELINE configs[4], NOT the headline fp32 metric]"
        dtype_out = {"f32_split": "f32 (fp32 tensors and accumulation; products from hi/lo fp16 pairs of the fp32 operands, three "
                                  "fp16 MFMAs per product group)",
                     "f32": "f32 (fp32 operands on v_mfma_f32_32x32x2_f32)", "f16": "f16"}[args.dtype]
        line = dict(metric=metric, value=round(total_images / elapsed, 2),
                    one_batch_in_flight_value=round(total_images / elapsed_serial, 2),      # the reference's loop (trainer/tester.py:39-44): like for like
                    unit="images/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True, scaling="weak",
                    vs_baseline=None, dtype=dtype_out, precision=args.dtype, plugin_default_precision=DEFAULT_PRECISION,
                    data="synthetic (two seeded input batches taken in turn)",
                    rccl_ranks=world if use_dist else 0,
                    per_rank_value=[round(B * args.steps / t, 2) for t in per_rank],       # images/s of every rank's own steps
                    scaling_efficiency=(round(total_images / elapsed / (world * solo["value"]), 4) if solo else None),
                    solo_reference=solo,      # rank 0 alone on this node, same binary, same steps (null at N = 1: `value` is it)
                    rank_ms_per_step=dict(min=round(min(per_rank) / args.steps * 1e3, 3), max=round(max(per_rank) / args.steps * 1e3, 3),
                                          one_batch_in_flight_min=round(min(per_rank_serial) / args.steps * 1e3, 3),
                                          one_batch_in_flight_max=round(max(per_rank_serial) / args.steps * 1e3, 3)),
                    weight_broadcast=dict(bytes=bc_stats.get("bytes"), ms=bc_stats.get("ms"), blobs=bc_stats.get("blobs"),
                                          gbs=(round(bc_stats["bytes"] / (bc_stats["ms"] * 1e-3) / 1e9, 1) if (use_dist and bc_stats.get("ms")) else None),
                                          pack_ms_per_rank=bc_stats.get("pack_ms_per_rank"), packed_bytes=bc_stats.get("packed_bytes"),
                                          blobs_identical_across_ranks=bc_stats.get("blobs_identical_across_ranks"),
                                          backend="rccl" if use_dist else "none (single rank: copied to the GPU and packed there)",
                                          note="rank 0's raw fp32 state_dict (weights + BatchNorm statistics) to every rank as one blob, "
                                               "once, before any timed region; every rank packs it on its own device (bit-identical "
                                               "blobs: a checksum of each is all-gathered)"),
                    config=dict(workload="OrienMaskYOLOFPNPlus forward + OrienMaskYOLOPostProcess, %d x [3,%d,%d] per GPU "
                                         "(BASELINE configs[2]); seeded random-init weights (seed %d, obj_bias %g, head_gain %g): "
                                         "%s heads (dense: >400 candidates pass conf_thresh per image, NMS, 100 masks per image; "
                                         "sparse: a few tens of detections per image).  Parity of exactly this workload: "
                                         "tests/test_hip_parity.py::test_bench_workload_bs32_detections (these weights' saturated heads tie "
                                         "hundreds of scores at exactly 1.0, so detections are compared as sets per tie group)"
                                         % (B, H, W, WEIGHT_SEED, obj_bias, head_gain, args.heads),
                                per_gpu_batch=B, image_size=[H, W], forward_streams=args.streams, batches_in_flight=args.in_flight, detections_per_image=round(sum(int(d_["bbox"].shape[0]) for d_ in dets) / B, 1),
                                parallelism="batch shard x%d, one RCCL weight broadcast, no collective in the step" % world),
                    one_batch_in_flight=dict(value=round(total_images / elapsed_serial, 2), ms_per_step=round(elapsed_serial / args.steps * 1e3, 3),
                                             note="the same K steps one batch at a time (the reference's loop: forward, postprocess, "
                                                  "host reads the counts, next batch); the roofline object's per-kernel events were "
                                                  "recorded in THIS region, where kernels do not overlap.  `value` keeps "
                                                  "batches_in_flight whole batches enqueued on alternating HIP streams"),
                    roofline=roofline)
        # whole-step utilisation over the time behind `value` (kernels of the batches in flight overlap: per-kernel figures above
        # come from the one-at-a-time region, these from the step time itself)
        step_s = elapsed / args.steps
        line["roofline"]["step"] = dict(
            ms=round(step_s * 1e3, 3), batches_in_flight=args.in_flight,
            executed_tflops=round(total_exec / step_s / 1e12, 2), executed_frac=round(total_exec / step_s / 1e12 / peak_tf, 4),
            hbm_algorithmic_gbs=round(total_bytes / step_s / 1e9, 1), hbm_algorithmic_frac=round(total_bytes / step_s / 1e9 / PEAK_HBM_GBS, 4),
            hbm_pmc_gbs=round(conv_stack["bytes_per_step"] / step_s / 1e9, 1) if conv_stack else None,
            hbm_pmc_frac=round(conv_stack["bytes_per_step"] / step_s / 1e9 / PEAK_HBM_GBS, 4) if conv_stack else None,
            note="forward flops the matrix pipe executed / algorithmic forward bytes / PMC conv-stack bytes, each per step, over the "
                 "step time of the timed region behind `value` (forward + postprocess, batches_in_flight batches overlapping)")
        if f32_operands is not None:
            line["f32_operands"] = f32_operands
        if f16_config is not None:
            line["f16_config"] = f16_config
        if small is not None:
            line["small_batches"] = small
        if not args.no_extras:
            line["extras"] = measure_neighbours(dev, dets, B)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, x_cpu, f16=f16)
        print(json.dumps(line))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
