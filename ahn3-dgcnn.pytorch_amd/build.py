"""Build recipe for libdgcnn_hip.so (explicit hipcc, gfx950 only, in-tree output).

hipcc cross-compiles without a GPU, so this runs in the CPU-only container; the built
.so is git-ignored but travels with the repo snapshot to the GPU box.

    python "ahn3-dgcnn.pytorch_amd/build.py" [--force] [--resources]
"""
import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libdgcnn_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-ffp-contract=off",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-function",
         # MFMA accumulators in VGPRs: the epilogues consume them with VALU instructions, and the default
         # AGPR form costs one v_accvgpr_read per accumulator register (64 per chunk in the fp16 streams)
         "-mllvm", "-amdgpu-mfma-vgpr-form"]
# no SLP packing of independent fp32 operations into v_pk_*_f32 in these files: beside MFMAs the packed forms are slower
# than the two scalar instructions they replace (edge_mlp_bwd16: 234 -> 225 us, edge_mlp_fwd: 129 -> 125 us)
FILE_FLAGS = {"edge_mlp_fwd16.hip": ["-fno-slp-vectorize"], "edge_mlp_bwd16.hip": ["-fno-slp-vectorize"]}
FLAGS += os.environ.get("DGCNN_EXTRA_HIPCC_FLAGS", "").split()  # experiments only (e.g. ablation -D switches)

def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "dgcnn_hip.h"))
    hdrs.append(os.path.abspath(__file__))
    return hdrs


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src, force, resources):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    stamp = obj + ".sha"
    dig = _digest([src] + _deps())
    if (not force and not resources and os.path.exists(obj) and os.path.exists(stamp)
            and open(stamp).read() == dig):
        return obj, ""
    cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
    if resources:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, p.stderr))
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, p.stderr


def _resource_table(stderr):
    rows, cur = [], None
    for line in stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for key in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize \\[bytes/lane\\]", "Occupancy \\[waves/SIMD\\]",
                    "LDS Size \\[bytes/block\\]"):
            m = re.search(r"remark:\s+" + key + r": (\d+)", line)
            if m and cur is not None:
                cur[key.split(" ")[0].replace("\\", "")] = int(m.group(1))
    out = []
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip() or r["name"]
        name = re.sub(r"\(.*", "", name)
        out.append("%-70s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %d lds %6d" % (
            name[:70], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1), r.get("ScratchSize", -1),
            r.get("Occupancy", -1), r.get("LDS", -1)))
    return "\n".join(out)


def build(force=False, resources=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, resources), srcs))
    objs = [o for o, _ in res]
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError("link failed:\n" + p.stderr)
        if verbose:
            print("linked", LIB)
    if resources:
        for _, err in res:
            t = _resource_table(err)
            if t:
                print(t)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, resources="--resources" in sys.argv, verbose=True)
