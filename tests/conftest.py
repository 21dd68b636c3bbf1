import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

ANCHORS_YOLOV4 = [[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]]
ANCHOR_MASK = [[6, 7, 8], [3, 4, 5], [0, 1, 2]]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def post_cfg(size_hw):
    """The reference's orienmask_yolo_coco_544_anchor4 postprocess constants
    (/root/reference/config/base.py:219-236) at an arbitrary /32 image size."""
    h, w = int(size_hw[0]), int(size_hw[1])
    return dict(grid_size=[[h // 32, w // 32], [h // 16, w // 16], [h // 8, w // 8]], image_size=[h, w],
                anchors=ANCHORS_YOLOV4, anchor_mask=ANCHOR_MASK, num_classes=80, conf_thresh=0.005,
                nms_pre=400, nms_post=100, orien_thresh=0.3)


@pytest.fixture(scope="session")
def built():
    """Build every native piece once per test session (HIP library + oracle C code)."""
    import __graft_entry__
    __graft_entry__.build()
    return True


def golden_files(prefix):
    return sorted(f for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def fixture_weights_and_input(g):
    """(state_dict, input) of a tests/golden/fwd_*.npz fixture, regenerated from its seeds (and, for the heavy-tailed stress
    fixtures, its stored per-layer normalisation factors) -- tools/gen_golden.py made them the same way."""
    from orienmask_amd import synth
    size = tuple(int(v) for v in g["size"]); batch = int(g["batch"])
    if "stress" in g.files:
        sd = synth.synth_state_dict_stress(int(g["wseed"]), g["norms"], obj_bias=float(g["obj_bias"]), head_gain=float(g["head_gain"]))
        x = synth.synth_image_batch_stress(int(g["xseed"]), batch, size[0], size[1])
    else:
        sd = synth.synth_state_dict(int(g["wseed"]), obj_bias=float(g["obj_bias"]), head_gain=float(g["head_gain"]))
        x = synth.synth_image_batch(int(g["xseed"]), batch, size[0], size[1])
    return sd, x
