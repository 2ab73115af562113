"""Compile the reference's own native extension (detectron2._C, CPU files only) from the
sources where they lie under /root/reference into oracle/_ref/ (git-ignored, travels to the
GPU box as a prebuilt .so).  No reference source is copied; nothing is stubbed: the csrc CPU
files build against the torch headers of this image unmodified
(reference setup.py:41-132 lists the same files; bindings csrc/vision.cpp:70-117).

Used to (1) pin oracle.c's ROIAlign restatement, (2) serve as `detectron2._C` inside
oracle/refshim.py, (3) time the reference ROIAlign as a cpu_baseline of kind "reference".
"""
import glob
import os
import sys

REF = "/root/reference/detectron2/layers/csrc"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def build(verbose=False):
    if not os.path.isdir(REF):
        print("reference tree absent; using prebuilt oracle/_ref if any")
        return None
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load

    srcs = [os.path.join(REF, "vision.cpp")]
    srcs += sorted(glob.glob(os.path.join(REF, "*", "*_cpu.cpp")))
    srcs += sorted(glob.glob(os.path.join(REF, "cocoeval", "*.cpp")))
    mod = load(
        name="_C",
        sources=srcs,
        extra_include_paths=[REF],
        extra_cflags=["-O2"],
        build_directory=OUT,
        with_cuda=False,
        verbose=verbose,
    )
    return mod


if __name__ == "__main__":
    sys.dont_write_bytecode = True
    m = build(verbose=True)
    print("built", m)
