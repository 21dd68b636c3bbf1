"""Compile the reference's own CPU NMS (eval/src/nms_cpu.cpp) into oracle/_ref/.

TEST INFRASTRUCTURE ONLY.  Runs only where /root/reference exists (the build container);
the GPU box uses the prebuilt oracle/_ref/*.so that travels with the snapshot.

The source is compiled from where it lies; nothing is copied into the repository.  torch
2.10 no longer accepts `AT_DISPATCH_FLOATING_TYPES(dets.type(), ...)` at nms_cpu.cpp:67
(the file's own comment there notes the alternative), so the translation unit is streamed
through `sed` on stdin with that one token changed to `dets.scalar_type()`; the algorithm
(lines 4-63) is compiled untouched.  No headers, libraries or tools are stubbed.
"""
import os
import subprocess
import sys
import sysconfig


def main(ref_root):
    import torch
    from torch.utils import cpp_extension
    here = os.path.dirname(os.path.abspath(__file__))
    out_dir = os.path.join(here, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(ref_root, "eval", "src", "nms_cpu.cpp")
    out = os.path.join(out_dir, "nms_cpu_ref" + sysconfig.get_config_var("EXT_SUFFIX"))
    if os.path.exists(out) and os.path.getmtime(out) > os.path.getmtime(src):
        print("oracle/_ref: up to date")
        return 0
    inc = []
    for p in cpp_extension.include_paths():
        inc += ["-isystem", p]
    inc += ["-isystem", sysconfig.get_paths()["include"]]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = ["g++", "-x", "c++", "-", "-O2", "-fPIC", "-shared", "-std=c++17", "-w",
           "-DTORCH_EXTENSION_NAME=nms_cpu_ref", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % abi] + inc + \
          ["-L" + libdir, "-Wl,-rpath," + libdir, "-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-o", out]
    sed = subprocess.Popen(["sed", "67s/dets\\.type()/dets.scalar_type()/", src], stdout=subprocess.PIPE)
    rc = subprocess.call(cmd, stdin=sed.stdout)
    sed.wait()
    if rc != 0:
        print("oracle/_ref: build failed (rc=%d)" % rc)
        return rc
    print("oracle/_ref: built", out)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference"))
