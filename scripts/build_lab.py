"""Builds scripts/lab_mmx.hip against the plugin's object files (the round-2 kernel comes from kernels_mmt.hip.o):
    python scripts/build_lab.py [--all-types]   ->  gpurun_out/lab_mmx   (gpurun_out/ travels to the GPU box? no: see below)
The binary is written to eagle-in-llama.cpp_amd/lib/ (git-ignored, shipped with the snapshot)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "eagle-in-llama.cpp_amd")
sys.path.insert(0, PKG)
import importlib.util
spec = importlib.util.spec_from_file_location("b", os.path.join(PKG, "build.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
b.build_plugin()
objs = [os.path.join(PKG, "lib", "obj-plugin", n + ".hip.o") for n in ("kernels_mmt", "kernels_mmvq", "kernels_mmq", "kernels_tile")]
out = os.path.join(PKG, "lib", "lab_mmx")
obj = "/tmp/lab_mmx.o"
flags = (["-DLAB_ALL_TYPES"] if "--all-types" in sys.argv else []) + (["-save-temps", "-Rpass-analysis=kernel-resource-usage"] if "--temps" in sys.argv else [])
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{PKG}/csrc", f"-I{ROOT}/scripts", "--offload-arch=gfx950", "-Wno-unused-result", "-x", "hip", "-c",
       os.path.join(ROOT, "scripts", "lab_mmx.hip"), "-o", obj] + flags
print(" ".join(cmd)); subprocess.check_call(cmd, cwd="/tmp")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", obj] + objs + ["-o", out]
print(" ".join(cmd)); subprocess.check_call(cmd, cwd="/tmp")
print(out)
