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


# ---------------------------------------------------------------------------------------------------------------------
# Test-session memoisation (the GPU suite had grown to 600 s of a 1200 s limit, most of it host work
# repeated per parametrisation).  Test infrastructure only -- the product packs every time it is asked to:
#   * synth.synth_state_dict(seed, ...): 1.8 s per call, ~40 calls with a dozen distinct arguments;
#   * pack.pack_state_dict / _split / _f16: 1-2 s each per model built from an already-seen state dict (the two precisions of a
#     fixture, the two models of an A/B test): keyed by a fingerprint of the CONTENT (every tensor's shape, its first elements
#     and its float64 sum), so a test that changes a weight gets a fresh pack.
# Small LRUs: a state dict is 255 MB, a set of packed blobs ~2 GB.  (Round 5 also moved the packing itself onto the device the
# weights are on -- pack.py -- which took most of the remaining cost away; the memo stays for the host-side packs.)
def _lru(cache, key, make, size):
    if key in cache:
        cache[key] = cache.pop(key)          # most recent last
        return cache[key]
    val = make()
    cache[key] = val
    while len(cache) > size:
        cache.pop(next(iter(cache)))
    return val


def _fingerprint(sd):
    """Content hash of a state dict: sha1 over every tensor's name, shape, dtype and BYTES (a perturbed weight with the same sum
    and the same first elements is another key -- the parity tests must validate the packer, not this cache)."""
    import hashlib
    import torch
    h = hashlib.sha1()
    for k, v in sd.items():
        h.update(repr(k).encode())
        if torch.is_tensor(v):
            t = v.detach().cpu().contiguous()
            h.update(repr((tuple(t.shape), str(t.dtype))).encode())
            h.update(t.reshape(-1).view(torch.uint8).numpy().tobytes() if t.numel() else b"")
        else:
            h.update(repr(v).encode())
    return h.hexdigest()


def _layers_key(layers):
    """Everything of the layer table a packer looks at (names, shapes, offsets): two tables of the same length and total size
    that differ in any of it are different keys."""
    import hashlib
    return hashlib.sha1(repr([sorted((k, repr(v)) for k, v in l.items()) if isinstance(l, dict) else repr(l) for l in layers]).encode()).hexdigest()


def _install_memo():
    from orienmask_amd import pack, synth
    if getattr(synth.synth_state_dict, "_memo", False):
        return
    sd_cache, blob_cache = {}, {}
    raw_synth = synth.synth_state_dict

    def synth_state_dict(*a, **kw):
        return dict(_lru(sd_cache, (a, tuple(sorted(kw.items()))), lambda: raw_synth(*a, **kw), 4))      # tensors shared: never written by tests

    synth_state_dict._memo = True
    synth.synth_state_dict = synth_state_dict
    for name in ("pack_state_dict", "pack_state_dict_split", "pack_state_dict_f16"):
        raw = getattr(pack, name)

        def packed(state_dict, layers, total, _raw=raw, _name=name):
            sd = pack.unwrap_checkpoint(state_dict)
            key = (_name, _fingerprint(sd), str(pack._blob_device(sd, layers)), int(total), _layers_key(layers))
            return _lru(blob_cache, key, lambda: _raw(state_dict, layers, total), 6).clone()

        packed.__wrapped__ = raw
        setattr(pack, name, packed)


_install_memo()


def golden_files(prefix):
    return sorted(f for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def fixture_weights_and_input(g):
    """(state_dict, input) of a tests/golden/fwd_*.npz fixture, regenerated from its seeds (and, for the heavy-tailed stress
    fixtures, its stored per-layer normalisation factors) -- tools/gen_golden.py made them the same way."""
    from orienmask_amd import synth
    size = tuple(int(v) for v in g["size"]); batch = int(g["batch"])
    if "trained" in g.files:        # converged-network statistics: the BatchNorm running statistics are stored in the fixture
        sd = synth.synth_state_dict_trained(int(g["wseed"]), (g["bn_mean"], g["bn_var"], g["head_norms"]),
                                            obj_bias=float(g["obj_bias"]), head_gain=float(g["head_gain"]))
        x = synth.synth_image_batch_stress(int(g["xseed"]), batch, size[0], size[1])
    elif "stress" in g.files:
        sd = synth.synth_state_dict_stress(int(g["wseed"]), g["norms"], obj_bias=float(g["obj_bias"]), head_gain=float(g["head_gain"]))
        x = synth.synth_image_batch_stress(int(g["xseed"]), batch, size[0], size[1])
    else:
        sd = synth.synth_state_dict(int(g["wseed"]), obj_bias=float(g["obj_bias"]), head_gain=float(g["head_gain"]))
        x = synth.synth_image_batch(int(g["xseed"]), batch, size[0], size[1])
    return sd, x
