"""Build recipe for the MI355X ggml backend plugin and its host-side harness (hipcc, gfx950 only).

    python eagle-in-llama.cpp_amd/build.py [--force]

Outputs (in-tree, git-ignored, shipped to the GPU box by gpurun):
    eagle-in-llama.cpp_amd/lib/libggml-mi355x.so     the plugin (C ABI: include/ggml_mi355x.h)
    eagle-in-llama.cpp_amd/lib/libeagle_host.so      C++ host: graph builders + speculative driver + C API
"""
import os, subprocess, sys, glob, hashlib, json

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "lib")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", f"-I{ROOT}/include", f"-I{HERE}/csrc",
          "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(p.encode()); h.update(open(p, "rb").read())
    return h.hexdigest()


def _run(cmd):
    print("  $", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _build_objs(srcs, objdir, flags, force):
    os.makedirs(objdir, exist_ok=True)
    hdrs = glob.glob(f"{HERE}/csrc/*.h") + glob.glob(f"{HERE}/host/*.h") + glob.glob(f"{ROOT}/include/*.h")
    stamp_path = os.path.join(objdir, "stamps.json")
    stamps = json.load(open(stamp_path)) if os.path.exists(stamp_path) else {}
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        d = _digest([s] + hdrs, " ".join(flags))
        objs.append(o)
        if not force and os.path.exists(o) and stamps.get(s) == d:
            continue
        cmd = [HIPCC] + flags + ["-c", s, "-o", o]
        print("  $", " ".join(cmd), flush=True)
        procs.append((s, d, subprocess.Popen(cmd)))
    for s, d, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"compile failed: {s}")
        stamps[s] = d
    json.dump(stamps, open(stamp_path, "w"))
    return objs, bool(procs)


def build_plugin(force=False, lab=False):
    """lab=True: the diagnostic variant (-DMI_LAB: A/B knobs read from the environment, phase-stamp kernel instantiations) as
    lib/libggml-mi355x-lab.so, for scripts/ only; the shipped plugin carries neither."""
    os.makedirs(LIB, exist_ok=True)
    srcs = sorted(glob.glob(f"{HERE}/csrc/*.hip") + glob.glob(f"{HERE}/csrc/*.cpp"))
    # -fno-strict-aliasing: kernels view LDS / registers through several types (f16 probabilities over an fp32 score row, packed int8 in int32)
    flags = COMMON + [f"--offload-arch={ARCH}", "-x", "hip", "-fno-strict-aliasing"] + (["-DMI_LAB=1"] if lab else [])
    objs, changed = _build_objs(srcs, os.path.join(LIB, "obj-plugin-lab" if lab else "obj-plugin"), flags, force)
    out = os.path.join(LIB, "libggml-mi355x-lab.so" if lab else "libggml-mi355x.so")
    if changed or not os.path.exists(out):
        _run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out] + objs + ["-ldl", "-lpthread"])
    return out


def build_host(force=False):
    os.makedirs(LIB, exist_ok=True)
    srcs = sorted(glob.glob(f"{HERE}/host/*.cpp"))
    if not srcs:
        return None
    flags = COMMON + [f"-I{HERE}/host"]
    os.makedirs(os.path.join(LIB, "obj-host"), exist_ok=True)
    objs, procs = [], []
    hdrs = glob.glob(f"{HERE}/host/*.h") + glob.glob(f"{ROOT}/include/*.h")
    stamp_path = os.path.join(LIB, "obj-host", "stamps.json")
    stamps = json.load(open(stamp_path)) if os.path.exists(stamp_path) else {}
    for s in srcs:
        o = os.path.join(LIB, "obj-host", os.path.basename(s) + ".o")
        d = _digest([s] + hdrs, " ".join(flags))
        objs.append(o)
        if not force and os.path.exists(o) and stamps.get(s) == d:
            continue
        cmd = ["g++"] + flags + ["-fopenmp", "-c", s, "-o", o]
        print("  $", " ".join(cmd), flush=True)
        procs.append((s, d, subprocess.Popen(cmd)))
    for s, d, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"compile failed: {s}")
        stamps[s] = d
    json.dump(stamps, open(stamp_path, "w"))
    out = os.path.join(LIB, "libeagle_host.so")
    if procs or not os.path.exists(out):
        _run(["g++", "-shared", "-fPIC", "-fopenmp", "-o", out] + objs + ["-ldl", "-lpthread"])
    return out


def build_all(force=False):
    return build_plugin(force), build_host(force)


if __name__ == "__main__":
    if "--lab" in sys.argv:
        print(build_plugin("--force" in sys.argv, lab=True))
    else:
        print(build_all("--force" in sys.argv))
