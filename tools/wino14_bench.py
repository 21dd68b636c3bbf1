"""Time conv_wino14.hip (the fused F(4,3) split-operand form) on the forward's stride-1 3x3 layer shapes at bs=32, 544x544, next to
the two-kernel F(2x4) split form on the same shapes:   gpurun -- 'python tools/wino14_bench.py'"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from orienmask_amd import lib as omlib  # noqa: E402
from orienmask_amd.pack import winograd14_weights_split, winograd_weights_split  # noqa: E402

SHAPES = [(272, 32, 64, 1), (136, 64, 128, 2), (68, 128, 256, 11), (34, 256, 512, 11), (17, 512, 1024, 7), (136, 128, 256, 5)]


def main():
    if os.environ.get("OM_LIB"):
        omlib.LIB_PATH = os.path.abspath(os.environ["OM_LIB"])
    L = omlib.load()
    dev = torch.device("cuda:0")
    B = 32
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    tot14 = tot24 = tot12 = 0.0
    shapes = SHAPES if not os.environ.get("OM_SHAPES") else [SHAPES[int(i)] for i in os.environ["OM_SHAPES"].split(",")]
    for hw, cin, cout, n in shapes:
        # OM_W14_PROBE_INPUT=1 (with a -DW14_BIG_PROBE=1 build, conv_wino14.hip): the input tensor allocated twice as large (pixel stride
        # 2 cin) -- the probe reads a transformed input of 1.5x the activation's bytes from it
        ps = 2 * cin if os.environ.get("OM_W14_PROBE_INPUT") else cin
        x = torch.randn(B, hw, hw, ps, device=dev)
        if os.environ.get("ZERO_X"):        # the same instruction stream on zeros: what the clock under load costs (profiles/r05_experiments.md section 4)
            x.zero_()
        w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
        out = torch.empty(B, hw, hw, cout, device=dev)
        hd = torch.zeros(cout, device=dev)
        u14, e14 = winograd14_weights_split(w, cout)
        u24, e24 = winograd_weights_split(w, cout)
        s14 = torch.pow(torch.tensor(2.0), -e14.float()).to(dev)
        s24 = torch.pow(torch.tensor(2.0), -e24.float()).to(dev)
        u14, u24 = u14.to(dev), u24.to(dev)
        scratch = torch.empty(L.om_conv2d_winograd24_scratch_bytes(B, hw, hw, cin), dtype=torch.uint8, device=dev)
        st = omlib.current_stream_ptr(dev)

        def run14_dual():
            L.om_set_wino14_variant(1); run14(); L.om_set_wino14_variant(0)

        def run14():
            omlib.check(L.om_conv2d_wino14_split(p(x), B, hw, hw, cin, ps, p(u14), p(s14), p(hd), cout, 1, None, 0, p(out), cout, None, st), "w14")

        def run24():
            omlib.check(L.om_conv2d_winograd24_split(p(x), B, hw, hw, cin, cin, p(u24), p(s24), p(hd), cout, 1, None, 0, p(out), cout,
                                                     p(scratch), scratch.numel(), None, st), "w24")
        wide_ok = cout % 128 == 0
        nb = L.om_conv2d_wino14_wide_scratch_bytes(B, hw, hw, cin)
        vscratch = torch.empty(nb if wide_ok else 16, dtype=torch.uint8, device=dev)

        def run14_wide():       # the two-kernel wide form (round 6): V pre-pass + 128 x 128 tiles
            omlib.check(L.om_conv2d_wino14_wide(p(x), B, hw, hw, cin, ps, p(u14), p(s14), p(hd), cout, 1, None, 0, p(out), cout, p(vscratch), nb, None, st), "w14 wide")
        res = []
        dual = bool(L.om_wino14_dual_built())       # only in libraries built with W14D=1
        if os.environ.get("OM_W14_WIDE"):           # first column: the wide form instead of the dual-role kernel
            for fn in ((run14_wide if wide_ok else run14), run14):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10):
                    fn()
                b.record()
                torch.cuda.synchronize()
                res.append(a.elapsed_time(b) / 10)
            print("%3dx%-3d %4d->%-4d x%2d  two-kernel wide form %.3f ms   fused twelve-wave %.3f ms" % (hw, hw, cin, cout, n, res[0], res[1]), flush=True)
            tot14 += n * res[0]; tot12 += n * res[1]
            continue
        for fn in ((run14_dual if dual else run14), run24, run14):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                fn()
            b.record()
            torch.cuda.synchronize()
            res.append(a.elapsed_time(b) / 10)
        fl = 2.0 * B * hw * hw * cin * cout * 9
        print("%3dx%-3d %4d->%-4d x%2d  fused F(4,3) dual-role %.3f ms (%.0f TF alg, %.0f TF executed)   twelve-wave %.3f ms   F(2x4) two kernels %.3f ms" % (
            hw, hw, cin, cout, n, res[0], fl / res[0] / 1e9, 1.5 * fl / res[0] / 1e9, res[2], res[1]), flush=True)
        tot14 += n * res[0]; tot24 += n * res[1]; tot12 += n * res[2]
    print("all 37 layers: fused dual-role %.2f ms, twelve-wave %.2f ms, two-kernel %.2f ms" % (tot14, tot12, tot24))


if __name__ == "__main__":
    main()
