"""Micro-benchmark of the HBM-bound kernels at cfg2 size with their algorithmic bytes.  python tools/hbm_bench.py [D H W]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from estdepth_amd.microbench import hbm_kernels_standalone

dims = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else [64, 120, 160]
for name, r in hbm_kernels_standalone(*dims).items():
    print("%-28s %8.1f us  %7.1f MB  %6.2f TB/s  (%.0f%% of 8 TB/s)" % (name, r["avg_launch_us"], r["algorithmic_mb_per_launch"],
                                                                    r["achieved_gbs"] / 1e3, r["frac"] * 100))
