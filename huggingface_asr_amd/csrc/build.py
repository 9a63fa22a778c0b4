"""Build libhfasr_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python huggingface_asr_amd/csrc/build.py [--force]

One object per .hip (compiled in parallel), linked into huggingface_asr_amd/libhfasr_hip.so (in-tree:
the .so is git-ignored but travels to the GPU box with the gpurun snapshot).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libhfasr_hip.so")
OBJ = os.path.join(HERE, "build")
SOURCES = ["gemm_bf16.hip", "gemm_glds.hip", "gemm_8p.hip", "norm.hip", "conv.hip", "attention.hip", "fbank.hip", "ctc.hip", "ctc_prefix.hip", "decoder.hip", "decoder_step.hip", "decoder_fused.hip", "beam_step.hip", "whisper.hip", "encoder.hip",
           "train_ops.hip", "gemm_tn.hip", "bgemm.hip", "attn_bwd.hip", "conv_bwd.hip", "loss_bwd.hip", "dropout.hip", "bestrq.hip", "specaug.hip", "speed.hip"]
# -packed-fp32-ops (device side only): no v_pk_{add,mul,fma}_f32 in any kernel.  A wave executing packed-f32 VALU ops next to the LDS-DMA GEMM's
# waves on one CU (kernels from two streams) lost the low-half product on 16 lanes in ~3 % of launches (tools/dbg/README.md); without packed
# ops the same pairing ran clean, and beside MFMAs they are slower than their scalar pairs anyway (MI355X_MICROARCH.md, fillers table).
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-result", "-munsafe-fp-atomics",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]      # (the host pass prints "not a recognized feature": harmless)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    deps = [os.path.join(HERE, src), os.path.join(HERE, "common.hpp"), os.path.join(HERE, "gemm_args.hpp"), os.path.join(os.path.dirname(PKG), "include", "hfasr_hip.h"),
            os.path.abspath(__file__)]                                   # the flags live in this file
    if _stale(obj, deps):
        subprocess.run(["hipcc", *FLAGS, "-c", os.path.join(HERE, src), "-o", obj], check=True)
    return obj


def build(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    for f in os.listdir(OBJ):          # objects of sources that no longer exist
        if f.endswith(".o") and f.replace(".o", ".hip") not in SOURCES:
            os.remove(os.path.join(OBJ, f))
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if force or _stale(OUT, objs):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs], check=True)
        check_no_packed_f32(objs)
    return OUT


def check_no_packed_f32(objs) -> None:
    """Fail the build if any gfx950 code object of the library contains packed-f32 VALU arithmetic (v_pk_{add,mul,fma}_f32): the flag above is passed
    through -Xclang and a toolchain that stopped honouring it would re-open the co-residency fault silently (DESIGN.md 'Concurrent kernels')."""
    import re
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    bundler, objdump, objcopy = (os.path.join(llvm, t) for t in ("clang-offload-bundler", "llvm-objdump", "llvm-objcopy"))
    if not all(os.path.exists(t) for t in (bundler, objdump, objcopy)):
        raise RuntimeError(f"cannot verify the packed-f32-free build: clang-offload-bundler / llvm-objdump / llvm-objcopy not found under {llvm}")
    pat = re.compile(r"\bv_pk_(add|mul|fma)_f32\b")
    with tempfile.TemporaryDirectory() as td:
        for o in objs:
            fb, co = os.path.join(td, os.path.basename(o) + ".fatbin"), os.path.join(td, os.path.basename(o) + ".co")
            subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", o, fb], check=True)        # the device code of a host object
            subprocess.run([bundler, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fb}", f"--output={co}"], check=True)
            dis = subprocess.run([objdump, "-d", "--mcpu=gfx950", co], check=True, capture_output=True, text=True).stdout
            hits = pat.findall(dis)
            if hits:
                os.remove(OUT)
                raise RuntimeError(f"{os.path.basename(o)}: {len(hits)} packed-f32 VALU instructions in the gfx950 code object; the library must be built without them")


if __name__ == "__main__":
    print(build("--force" in sys.argv))
