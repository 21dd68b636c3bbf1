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
    dev = torch.device("cuda", 0)
    out = {}
    for tag, (wseed, bias, gain) in {"bench weights": (3, -16.0, 4.0), "unsaturated heads": (3, -3.0, 0.7)}.items():
        sd = synth.synth_state_dict(wseed, obj_bias=bias, head_gain=gain)
        x = synth.synth_image_batch(1000, 6, 544, 544)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        truth = R.forward(sd64, x[:2].double())
        ref32 = R.forward(sd, x[:2])
        rec = {"reference fp32 (torch CPU)": dict(max=max(max(rel(a, c), rel(b, d)) for (a, b), (c, d) in zip(ref32, truth)),
                                                   rms=max(max(rms(a, c), rms(b, d)) for (a, b), (c, d) in zip(ref32, truth)))}
        for prec in ("f32", "f32_split"):
            net = OrienMaskYOLOFPNPlus(3, 80).eval().set_precision(prec)
            net.load_state_dict(sd, strict=True)
            net = net.to(dev)
            with torch.no_grad():
                got = net(x.to(dev))
            torch.cuda.synchronize()
            rec["HIP " + prec] = dict(max=max(max(rel(a[:2].cpu(), c), rel(b[:2].cpu(), d)) for (a, b), (c, d) in zip(got, truth)),
                                      rms=max(max(rms(a[:2].cpu(), c), rms(b[:2].cpu(), d)) for (a, b), (c, d) in zip(got, truth)))
        out[tag] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
