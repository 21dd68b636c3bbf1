"""Error of the head tensors against a float64 evaluation of the same network, for
  * the reference's own fp32 arithmetic (torch CPU, oracle/orienmask_ref.py:forward),
  * the HIP forward with fp32 operands (precision 'f32'),
  * the HIP forward with split operands in the F(2x4) Winograd GEMMs (precision 'f32_split').
Two 544x544 images inside a batch of 6 (so that the F(2x4) kernels run).  Prints one JSON object.
Test infrastructure: imports the oracle as the checker (tools/ is not shipped)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import orienmask_ref as R          # noqa: E402
from orienmask_amd import synth                # noqa: E402
from orienmask_amd.model import OrienMaskYOLOFPNPlus   # noqa: E402


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


def rms(a, b):
    return float(((a.double() - b) ** 2).mean().sqrt() / b.abs().max())


def main():
    import numpy as np
    dev = torch.device("cuda", 0)
    out = {}
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "fwd_f160x128_b1.npz"))
    cases = {"bench weights, 544x544, batch of 6": (3, -16.0, 4.0, 1000, 6, 544, 544),
             "fixture fwd_f160x128_b1 (one 160x128 image)": (int(g["wseed"]), float(g["obj_bias"]), float(g["head_gain"]), int(g["xseed"]),
                                                            1, int(g["size"][0]), int(g["size"][1]))}
    for tag, (wseed, bias, gain, xseed, nb, hh, ww) in cases.items():
        sd = synth.synth_state_dict(wseed, obj_bias=bias, head_gain=gain)
        x = synth.synth_image_batch(xseed, nb, hh, ww)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        nk = min(2, nb)
        truth = R.forward(sd64, x[:nk].double())
        ref32 = R.forward(sd, x[:nk])
        rec = {"reference fp32 (torch CPU)": dict(max=max(max(rel(a, c), rel(b, d)) for (a, b), (c, d) in zip(ref32, truth)),
                                                   rms=max(max(rms(a, c), rms(b, d)) for (a, b), (c, d) in zip(ref32, truth)))}
        for prec in ("f32", "f32_split"):
            net = OrienMaskYOLOFPNPlus(3, 80).eval().set_precision(prec)
            net.load_state_dict(sd, strict=True)
            net = net.to(dev)
            with torch.no_grad():
                got = net(x.to(dev))
            torch.cuda.synchronize()
            rec["HIP " + prec] = dict(max=max(max(rel(a[:nk].cpu(), c), rel(b[:nk].cpu(), d)) for (a, b), (c, d) in zip(got, truth)),
                                      rms=max(max(rms(a[:nk].cpu(), c), rms(b[:nk].cpu(), d)) for (a, b), (c, d) in zip(got, truth)))
        out[tag] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
