"""Static scan of gfx950 listings (`hipcc -S --cuda-device-only`) for scalar-memory loads INSIDE loops: kernel-argument fields hipcc
re-loads (s_load + s_waitcnt lgkmcnt(0)) where they are used instead of keeping them in SGPRs (round 6: up to 34 per iteration of a strip
loop; common.h: pc_pin is the fix).    python3 tools/isa_scalar_loads.py [file.hip ...]   (default: every kernel source)"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VGPR_FORM = {"conv3x3.hip", "conv3x3_bwd.hip", "up_bwd.hip", "convt2x2.hip"}


def listing(src):
    out = os.path.join(tempfile.gettempdir(), "isa_" + os.path.basename(src)[:-4] + ".s")
    extra = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] if os.path.basename(src) in VGPR_FORM else []
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include")] + extra +
                   ["-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    return out


def scan(path):
    filt = shutil.which("c++filt")
    rows, name, cur = [], None, []
    for l in open(path):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, cur = m.group(1), []
            continue
        if name is None:
            continue
        cur.append(l)
        if "s_endpgm" in l:
            in_loop = [x for x in cur if re.search(r"\ts_load", x) is not None]
            # a load is "in a loop" when the nearest preceding block label carries a loop annotation
            n, lab = 0, ""
            for x in cur:
                mm = re.match(r"^\.LBB\S+:\s*;\s*(.*)$", x) or re.match(r"^; %bb\.\d+:\s*;\s*(.*)$", x)
                if mm:
                    lab = mm.group(1)
                elif re.match(r"^\.LBB\S+:", x) or re.match(r"^; %bb\.\d+:", x):
                    lab = ""
                if "\ts_load" in x and "Loop" in lab:
                    n += 1
            if n:
                d = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() if filt else name
                rows.append((n, len(in_loop), d.replace("(anonymous namespace)::", "")))
            name = None
    return rows


def main():
    srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "popcorn_amd", "csrc", "*.hip")))
    for src in srcs:
        for n, tot, d in sorted(scan(listing(src)), reverse=True):
            print(f"{os.path.basename(src):20s} in-loop s_load {n:3d} (of {tot:3d})  {d[:120]}")


if __name__ == "__main__":
    main()
