"""ONE table of every ESTD_* switch in the tree: run-time environment variables (read by the Python host layer, bench.py and the C launchers)
and compile-time macros of the kernels (``#ifndef ESTD_X / #define ESTD_X default`` A/B and ablation switches).  Writes SWITCHES.md;
tests/test_switch_table.py fails when a switch appears in the code that the committed table does not list.
    python tools/switch_table.py            (rewrites SWITCHES.md)"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# what each RUN-TIME switch does and where its A/B stands (compile-time macros carry their own comment in the source)
NOTES = {
    "ESTD_LIB": "path of libestd_hip.so (a variant library built with ESTD_LIB_SUFFIX; ctypes binding)",
    "ESTD_LIB_SUFFIX": "build.py: suffix of the library / object directory (variant builds beside the default one)",
    "ESTD_BUILD_AB": "build.py: also compile / export / bind the superseded A/B kernels (depth-only and row-only Winograd, bf16 operand splits, operand-reuse wino2x)",
    "ESTD_BUILD_DEFS": "build.py: extra -D flags for every kernel file (compile-time switches below)",
    "ESTD_BINDING": "torch (TORCH_LIBRARY operators, default) | ctypes (the torch-free C ABI): both tested",
    "ESTD_CONV3D_ARITH": "f32 (default) | bf16x3 (operand split; ESTD_BUILD_AB builds)",
    "ESTD_CONV2D_ARITH": "f32 (default) | bf16x3 (operand split; ESTD_BUILD_AB builds)",
    "ESTD_CONV3D_ALGO": "wino2 (default: Winograd kernels, see ESTD_W3) | direct (27-tap implicit GEMM, the fallback of every instance) | wino (depth-only; ESTD_BUILD_AB builds)",
    "ESTD_CONV2D_ALGO": "wino2 (default: F(2x2,3x3)) | direct | wino (row-only; ESTD_BUILD_AB builds)",
    "ESTD_CONV2D_NT": "auto | 2 | 4: output-channel tiles per work item of the direct / row-only 2D kernels",
    "ESTD_W3": "1 (default): 32-output-channel 3x3x3 convolutions on the three-axis Winograd kernel; 0: two-axis kernel (the r5 A/B: 0.65 vs 0.81 ms)",
    "ESTD_W3_EXTRA": "1 (default): the key||value convolution (33 -> 32) on the three-axis kernel too; 0: two-axis kernel (0.73 vs 0.85 ms)",
    "ESTD_W3_XOUT": "1 (default, round 6): dres2 (33 -> 33) = 33 -> 32 on the three-axis kernel + output channel 32 as a pass of its own (csrc/conv3d_xout.hip); 0: the two-axis kernel's 33 -> 33 instance (1.00 vs 0.88 ms)",
    "ESTD_W2_XOUT": "1 (default): dres2 (33 -> 33) on the two-axis kernel's XOUT instance; 0: depth-only kernel (ESTD_BUILD_AB builds) or direct",
    "ESTD_W2X": "0 (default) | 1: plain 32 -> 32 on the operand-reuse two-axis kernel (ESTD_BUILD_AB builds; measured dead end, r5)",
    "ESTD_C2W2_DIL2": "1 (default): dilation-2 3x3 convolutions on the F(2x2,3x3) kernel; 0: direct kernel",
    "ESTD_GATE_IN_CONV": "1 (default): ConvGRU reset gate folded into the output convolution's loads; 0: separate gru_reset_apply pass (+0.15 ms / step)",
    "ESTD_FAST_PATH": "1 (default): DepthNetHybrid.accelerate() on a ROCm device; 0: plain module path (NCHW, library 2D networks)",
    "ESTD_FUSED_NORM": "1 (default): image normalisation + NHWC conversion in one kernel",
    "ESTD_MIX_GEMM": "1 (default): pre0 pushed in front of the warp as a channel mix of the 2D features",
    "ESTD_MIX_HIP": "1 (default): that mix on csrc/conv1x1.hip; 0: library GEMM (neutral)",
    "ESTD_R50_HIP": "1 (default): ResNet branch on the in-house kernels; 0: library convolutions",
    "ESTD_HIP_REFINE": "1 (default): 2D refinement glue on csrc/refine2d.hip",
    "ESTD_HIP_TO16": "1 (default): the two full-resolution 16-channel ConvBlocks on conv2d_k3_to16",
    "ESTD_HIP_1X1": "all (default) | auto | 0: which ResNet 1x1 convolutions run on csrc/conv1x1.hip (the step times the same either way)",
    "ESTD_HIP_TAPS": "1 (default): stride-2 3x3 / small-map 3x3 convolutions on csrc/conv2d_taps.hip; 0: library",
    "ESTD_HIP_STEM": "1 (default): PSM 3x3 stride-2 stem in-house",
    "ESTD_HIP_STEM7": "1 (default): ResNet 7x7 stride-2 stem in-house",
    "ESTD_HIP_POOL": "1 (default): max / average pooling in-house",
    "ESTD_HIP_SMALL_CONVS": "1 (default): PSM 1x1 / stride-2 convolutions on conv2d_small",
    "ESTD_HIP3X3_MIN_ITEMS": "128: fewest work items for which a 3x3 convolution takes the persistent Winograd kernel",
    "ESTD_GEMM_EPILOGUE": "1 (default): library GEMM with fused bias / ReLU where a 1x1 convolution is left to the library (ESTD_HIP_1X1 != all)",
    "ESTD_SPP_FUSED": "1 (default): SPP upsample + concatenation in one pass",
    "ESTD_OVERLAP_HEADS": "0 (default) | 1: stereo heads on a side stream (lost against the batched head launch, r3)",
    "ESTD_BATCH_HEAD1": "1 (default): stereo_head1 of all targets in one launch behind the fusion loop",
    "ESTD_GRAPH_SHARE_POOL": "1 (default): one graph memory pool per call shape; 0: one per capture (tens of GB at cfg5 size)",
    "ESTD_AG_ALGO": "auto (default) | collective | direct: memory-bank exchange (estdepth_amd/parallel.py)",
    "ESTD_C1X1_CFG": "force one tile configuration of csrc/conv1x1.hip (tools/conv1x1_cfg_sweep.py); latched at the first call",
    "ESTD_C1X1_LDS": "1 (default): LDS-tiled form of the 1x1 convolution where the dispatcher picks it; 0: direct form only",
    "ESTD_WA_BUF": "force buffer-load (1) / pointer (0) gathers in warp_attention for every source count",
    "ESTD_WA_OCC": "cap the resident workgroups per CU of warp_attention (sweep: profiles/r4_warp_attention_buf.txt)",
    "ESTD_FORCE_DIST": "bench.py: world-size-1 RCCL communicator on one GPU (code-path + overlap cost)",
    "ESTD_DIST_BACKEND": "bench.py: nccl (RCCL, default) | gloo",
    "ESTD_OVERSUBSCRIBED": "bench.py (internal): ranks share fewer GPUs than ranks (code-path run)",
    "ESTD_FORCE_DEVICE": "bench.py: device ordinal override",
    "ESTD_RESERVED_CUS": "bench.py: CUs the persistent grids leave free for the overlapped exchange (8)",
    "ESTD_RESERVE_SCOPE": "bench.py: auto | A | AB: in which stage the reserve holds",
    "ESTD_RESERVE_COST": "bench.py: 0 skips the third timed loop (no reserve) of an N > 1 run",
    "ESTD_RCCL_DEBUG": "bench.py: 0 = do not collect RCCL's INFO log into the line",
    "ESTD_AG_DIAG": "bench.py: 0 = do not time the other exchange algorithm",
    "ESTD_AG_DIAG_TIMEOUT": "bench.py: watchdog of that diagnostic (60 s)",
    "ESTD_AG_DIAG_TEST_HANG": "bench.py: test hook of the watchdog",
    "ESTD_AG_AUTO_RETIME": "bench.py: 0 = do not re-time on the chosen exchange algorithm",
    "ESTD_BENCH_CHILD": "bench.py (internal): the traced child run of replay_profile()",
    "ESTD_GRAPH_MEMORY": "bench.py: zero-copy (default) | copy",
    "ESTD_PIPELINE": "bench.py: off (default) | on: GraphedForward(pipeline=True), stage A of step k + 1 beside the second half of stage B of step k (not a robust gain: profiles/r6_pipeline_ab.txt)",
    "ESTD_PIPE_RELEASE": "pipeline mode A/B: mid (default: the next stage A waits for the previous call's 3D-convolution chain) | start (it does not: slower)",
    "ESTD_SUSTAINED_S": "bench.py: seconds of the sustained loop (20; 0 = skip)",
    "ESTD_CPU_LEG_CPUS": "bench.py (internal): cpu list of the pinned CPU-baseline child",
    "ESTD_NCHW_2D": "bench.py A/B: 1 = plain NCHW library 2D networks",
    "ESTD_PSM": "bench.py A/B: hip (default) | lib: PSM 3x3 convolutions on the library",
    "ESTD_FUSE_BN": "bench.py A/B: 0 = separate BN / add / ReLU passes",
    "ESTD_OVERLAP": "bench.py A/B: 0 = semantic branch on the main stream",
    "ESTD_C2_GRID_MULT": "row-only 2D Winograd kernel (ESTD_BUILD_AB builds): persistent-grid multiplier (A/B)",
    "ESTD_CTAPS_CFG": "force one block configuration of csrc/conv2d_taps.hip (100 TM + 10 TN + SK), latched at the first call",
    "ESTD_GRU_FAST": "1: sigmoid / tanh of the GRU kernels on v_exp / v_rcp (neutral, r4); default: libm forms",
    "ESTD_SWEEP_LINEAR": "set: linear block order in the plane sweep instead of the XCD-contiguous bricks (A/B, r2)",
    "ESTD_WINO2_WAVES": "4: four-wave form of the two-axis 3D Winograd kernel (r3 A/B); default 8 waves",
    "ESTD_COLLECT_LINES": "tools/collect_profiles.sh: 0 = stop before the bench lines",
}


def scan():
    env, macros = {}, {}
    py = glob.glob(os.path.join(ROOT, "estdepth_amd", "*.py")) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for f in sorted(py):
        for n, line in enumerate(open(f), 1):
            for m in re.finditer(r"environ(?:\.get|\.setdefault)?[\(\[]\s*\"(ESTD_[A-Z0-9_]+)\"(?:\s*,\s*([^)]+?))?[\)\]]", line):
                name, default = m.group(1), (m.group(2) or "").strip().strip('"')
                env.setdefault(name, [default if "os.path" not in default else "(path)", []])[1].append("%s:%d" % (os.path.relpath(f, ROOT), n))
            for m in re.finditer(r"\"(ESTD_[A-Z0-9_]+)\" in os\.environ|ESTD_[A-Z0-9_]+=", line):
                if m.group(1):
                    env.setdefault(m.group(1), ["", []])[1].append("%s:%d" % (os.path.relpath(f, ROOT), n))
    for f in sorted(glob.glob(os.path.join(ROOT, "estdepth_amd", "csrc", "*"))):
        for n, line in enumerate(open(f, errors="replace"), 1):
            m = re.search(r"getenv\(\"(ESTD_[A-Z0-9_]+)\"\)", line)
            if m:
                env.setdefault(m.group(1), ["", []])[1].append("%s:%d" % (os.path.relpath(f, ROOT), n))
            m = re.match(r"\s*#define (ESTD_[A-Z0-9_]+)\s+(\S+)\s*(?://\s*(.*))?$", line)
            if m and not m.group(1).endswith("_H") and "(" not in m.group(1) and m.group(1) not in ("ESTD_NO_FLOOR",) and not re.match(r"ESTD_(OK|ERR_|MAX_|LAUNCH)", m.group(1)):
                macros.setdefault((os.path.basename(f), m.group(1)), [m.group(2), n, (m.group(3) or "").strip()])
    return env, macros


def render():
    env, macros = scan()
    out = ["# ESTD_* switches (generated by `python tools/switch_table.py`; `tests/test_switch_table.py` keeps it complete)", "",
           "Defaults are what `DepthNetHybrid(...).cuda().eval()`, `bench.py` and the driver run.  Everything else is an A/B or ablation switch kept so that",
           "a recorded measurement can be repeated; switches whose alternative needs a superseded kernel say `ESTD_BUILD_AB builds`.", "",
           "## Run-time (environment variables)", "", "| switch | default | read at | what it selects |", "|---|---|---|---|"]
    for name in sorted(env):
        default, where = env[name]
        out.append("| `%s` | %s | %s | %s |" % (name, ("`%s`" % default) if default else "—", ", ".join(sorted(set(where))[:3]), NOTES.get(name, "")))
    out += ["", "## Compile-time (kernel macros: `ESTD_BUILD_DEFS=\"-DESTD_X=v\" ESTD_LIB_SUFFIX=_v python -m estdepth_amd.build`, then `ESTD_LIB=...`)", "",
            "`*ABL` macros are timing ablations (results are wrong when non-zero).  Files marked (A/B) are compiled with `ESTD_BUILD_AB=1` only.", "",
            "| file | macro | default | source comment |", "|---|---|---|---|"]
    ab = {"conv3d_wino.hip", "conv3d_split_bf16.hip", "conv2d_wino.hip", "conv2d_split_bf16.hip", "conv3d_wino2x.hip"}
    for (f, name) in sorted(macros):
        default, line, comment = macros[(f, name)]
        out.append("| %s%s:%d | `%s` | `%s` | %s |" % (f, " (A/B)" if f in ab else "", line, name, default, comment.replace("|", "/")[:160]))
    return "\n".join(out) + "\n", env, macros


if __name__ == "__main__":
    text, env, macros = render()
    missing = sorted(n for n in env if n not in NOTES)
    if missing:
        print("no description for: " + ", ".join(missing), file=sys.stderr)
    open(os.path.join(ROOT, "SWITCHES.md"), "w").write(text)
    print("SWITCHES.md: %d run-time switches, %d compile-time macros" % (len(env), len(macros)))
