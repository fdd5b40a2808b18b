"""Register / scratch account of the convolution kernels from the compiler itself: `-Rpass-analysis=kernel-resource-usage` per instance and, from
the ISA (`-S`), WHERE the scratch accesses of an instance with spilled registers sit -- inside a loop that issues MFMAs (the persistent tile loop:
the tap loops themselves are fully unrolled, so this is a per-TILE cost of that many dword accesses beside some thousand instructions) or in
straight-line code outside (set-up: once per launch).   python tools/kernel_resources.py [file.hip ...] > profiles/r6_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "estdepth_amd", "csrc")
FILES = sys.argv[1:] or ["conv3d_wino3.hip", "conv3d_xout.hip", "conv3d_wino2.hip", "conv3d_wino2_c16.hip", "conv2d_wino2.hip", "conv1x1.hip", "est_fusion.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


for f in FILES:
    src = os.path.join(CSRC, f)
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(tmp, "a.o")], capture_output=True, text=True)
        rec, cur = {}, None
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = rec.setdefault(m.group(1), {})
            for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"):
                m = re.search(re.escape(key) + r": (\d+)", line)
                if m and cur is not None and " " + key + ":" in line:
                    cur[key] = int(m.group(1))
        asm = os.path.join(tmp, "a.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", src, "-o", asm], capture_output=True, text=True)
        text = open(asm).read()
    names = demangle(list(rec))
    print("# %s" % f)
    print("%-78s %5s %5s %8s %6s %6s %5s   scratch accesses: per iteration of a loop with MFMAs (= per tile: the tap loops are unrolled) / per launch" % ("kernel instance", "VGPR", "AGPR", "scratch", "vspill", "sspill", "occ"))
    for mangled, d in rec.items():
        m = re.search(r"\n%s:.*?\n(.*?)\n\.Lfunc_end" % re.escape(mangled), text, flags=re.S)
        inside = outside = 0
        if m:
            lines = m.group(1).splitlines()
            labels = {mm.group(1): i for i, l in enumerate(lines) for mm in [re.match(r"(\.LBB\d+_\d+):", l)] if mm}
            loops = []
            for i, l in enumerate(lines):
                mm = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
                if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                    a = labels[mm.group(1)]
                    if any("v_mfma" in x for x in lines[a:i]):
                        loops.append((a, i))
            for i, l in enumerate(lines):
                if re.search(r"scratch_(load|store)|buffer_(load|store)_dword.*offen.*Spill|Folded (Spill|Reload)", l) and ("scratch_" in l or "Folded" in l):
                    if any(a <= i <= b for a, b in loops):
                        inside += 1
                    else:
                        outside += 1
        short = re.sub(r"\(anonymous namespace\)::|\(estd_\w+desc.*$|void ", "", names.get(mangled, mangled))
        print("%-78s %5d %5d %8d %6d %6d %5d   %d / %d" % (short[:78], d.get("VGPRs", -1), d.get("AGPRs", 0), d.get("ScratchSize [bytes/lane]", 0), d.get("VGPRs Spill", 0),
                                                     d.get("SGPRs Spill", 0), d.get("Occupancy [waves/SIMD]", 0), inside, outside))
    print()
