"""profiles/<round>_conv3d_pmc.json (what bench.py reads for roofline.traffic) from the three FETCH_SIZE / WRITE_SIZE passes that
tools/collect_profiles.sh leaves as <round>_conv3d_{wino2,wino,direct}_pmc.csv.      python tools/pmc_json.py [dir] [round prefix, default r4]"""
import csv, json, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
RND = sys.argv[2] if len(sys.argv) > 2 else "r5"
ALG = 64 * 120 * 160 * (32 + 32) * 4                      # read 32 + write 32 channels of fp32 per voxel
out = {"how": "tools/pmc_collect.sh \"FETCH_SIZE WRITE_SIZE\" -- python tools/conv_bench.py 3 10 (separate --pmc passes; values in KiB; gfx950: "
              "FETCH_SIZE counts 64-byte units as 32 -> x2, MI355X_MICROARCH.md); 3 volumes of 64x120x160 per launch",
       "algorithmic_bytes_per_volume": ALG, "hbm_bytes_per_volume_by_algo": {}, "detail": {}}
for algo, pat in (("wino3", "conv3d_wino3_kernel"), ("wino2", "conv3d_wino2_kernel"), ("wino", "conv3d_wino_kernel"), ("direct", "conv3d_k3_kernel")):
    if not os.path.exists(os.path.join(d, "%s_conv3d_%s_pmc.csv" % (RND, algo))):      # (the depth-only kernel: ESTD_BUILD_AB=1 builds only)
        continue
    with open(os.path.join(d, "%s_conv3d_%s_pmc.csv" % (RND, algo))) as f:
        rows = [r for r in csv.DictReader(f) if pat in r["kernel"]]
    r = max(rows, key=lambda r: float(r["launches"]))
    fetch, write = float(r["avg_FETCH_SIZE"]), float(r["avg_WRITE_SIZE"])
    per_launch = int((2 * fetch + write) * 1024)
    out["hbm_bytes_per_volume_by_algo"][algo] = per_launch // 3
    out["detail"][algo] = {"kernel": r["kernel"][:80], "fetch_kib": fetch, "write_kib": write, "hbm_bytes_per_launch": per_launch,
                           "ratio_to_algorithmic": round(per_launch / (3.0 * ALG), 3)}
with open(os.path.join(d, "%s_conv3d_pmc.json" % RND), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out["hbm_bytes_per_volume_by_algo"]), {k: v["ratio_to_algorithmic"] for k, v in out["detail"].items()})
