"""Every tile configuration of csrc/conv1x1.hip on the 16 ResNet-50 1x1 shapes of the semantic branch (3 images of 480x640): error against
fp64 and time per launch at sustained clocks, one child process per configuration (ESTD_C1X1_CFG is latched at the first call), the
library (hipBLASLt GEMM + epilogue) beside it.   python tools/conv1x1_cfg_sweep.py [cfg ...]      (default: every configuration)
   direct form: 100 TM + 10 TN + SK (441 421 221 444 244 224 ...);  LDS-tiled form: 1000 + 100 (BM / 32) + 10 (BN / 32) + U"""
import json
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 3
SHAPES = [  # name, H, W, cin, cout, stride, residual
    ("l1 conv1 64->64", 120, 160, 64, 64, 1, False), ("l1 conv3 64->256 +res", 120, 160, 64, 256, 1, True), ("l1 ds 64->256", 120, 160, 64, 256, 1, False),
    ("l1 conv1 256->64", 120, 160, 256, 64, 1, False), ("l2 conv1 256->128", 120, 160, 256, 128, 1, False), ("l2 conv3 128->512 +res", 60, 80, 128, 512, 1, True),
    ("l2 ds 256->512 s2", 120, 160, 256, 512, 2, False), ("l2 conv1 512->128", 60, 80, 512, 128, 1, False), ("l3 conv1 512->256", 60, 80, 512, 256, 1, False),
    ("l3 conv3 256->1024 +res", 30, 40, 256, 1024, 1, True), ("l3 ds 512->1024 s2", 60, 80, 512, 1024, 2, False), ("l3 conv1 1024->256", 30, 40, 1024, 256, 1, False),
    ("l4 conv1 1024->512", 30, 40, 1024, 512, 1, False), ("l4 conv3 512->2048 +res", 15, 20, 512, 2048, 1, True), ("l4 ds 1024->2048 s2", 30, 40, 1024, 2048, 2, False),
    ("l4 conv1 2048->512", 15, 20, 2048, 512, 1, False)]
CFGS = ["0", "lib", "441", "421", "221", "444", "244", "224", "1441", "1442", "1421", "1422", "1241", "1242", "1221", "1222", "1224", "1122", "1124"]


def child(cfg):
    import torch
    from estdepth_amd import ops
    from estdepth_amd.microbench import warm
    dev = "cuda"

    def t(f, n=40):
        warm(f, 0.08)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / n
    res = {}
    for name, H, W, cin, cout, s, has_res in SHAPES:
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(N, H, W, cin, device=dev, generator=g)
        w = torch.randn(cout, cin, device=dev, generator=g) / cin ** 0.5
        sc, sh = torch.rand(cout, device=dev, generator=g) + 0.5, torch.randn(cout, device=dev, generator=g)
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        r = torch.randn(N, Ho, Wo, cout, device=dev, generator=g) if has_res else None
        if cfg == "lib":
            wt = (w * sc[:, None]).t().contiguous()

            def lib():
                xs = x[:, ::s, ::s].contiguous() if s > 1 else x
                x2 = xs.reshape(-1, cin)
                if r is None:
                    return torch._addmm_activation(sh, x2, wt, use_gelu=False)
                y = torch.mm(x2, w.t()).view(N, Ho, Wo, cout).permute(0, 3, 1, 2)
                return ops.bn_act_nhwc_(y, sc, sh, True, r.permute(0, 3, 1, 2))
            res[name] = [t(lib), 0.0]
            continue
        try:
            out = ops.conv1x1_nhwc(x, w, sc, sh, s, True, r)
        except RuntimeError as e:
            res[name] = [None, str(e)[:60]]
            continue
        ref = torch.einsum("nhwc,oc->nhwo", x[:, ::s, ::s].double(), w.double()) * sc.double() + sh.double()
        if r is not None:
            ref = ref + r.double()
        ref = ref.clamp_min(0)
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        res[name] = [t(lambda: ops.conv1x1_nhwc(x, w, sc, sh, s, True, r)), err]
    print("RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    cfgs = sys.argv[1:] or CFGS
    table = {}
    for cfg in cfgs:
        env = dict(os.environ)
        if cfg not in ("lib",):
            env["ESTD_C1X1_CFG"] = cfg
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", cfg], env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        table[cfg] = json.loads(line[-1][7:]) if line else {"error": (r.stderr or "")[-300:]}
        if not line:
            print("cfg %s failed: %s" % (cfg, (r.stderr or "")[-300:]))
    print("%-26s" % "us per launch" + "".join("%8s" % c for c in cfgs) + "   best (x library)")
    for name, H, W, cin, cout, s, _ in SHAPES:
        row, best = "%-26s" % name, None
        for c in cfgs:
            v = table[c].get(name, [None, ""]) if "error" not in table[c] else [None, ""]
            bad = v[0] is None or (isinstance(v[1], float) and v[1] > 2e-6)
            row += "%8s" % ("-" if v[0] is None else ("%.1f%s" % (v[0], "!" if bad else "")))
            if not bad and c not in ("lib", "0") and (best is None or v[0] < best[0]):
                best = (v[0], c)
        lib = table.get("lib", {}).get(name, [None])[0] if "lib" in table else None
        gf = 2.0 * N * ((H - 1) // s + 1) * ((W - 1) // s + 1) * cin * cout / 1e9
        if best:
            row += "   %s %.1f us = %.0f TF/s" % (best[1], best[0], gf * 1e3 / best[0]) + ((" (x%.2f)" % (lib / best[0])) if lib else "")
        print(row)
    worst = max((v[1] for c in cfgs if c != "lib" and "error" not in table[c] for v in table[c].values() if isinstance(v[1], float)), default=0.0)
    print("largest max|err| / max|ref| over every configuration and shape: %.3g   ('!' = above 2e-6, '-' = configuration refuses the shape)" % worst)
