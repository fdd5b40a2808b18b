"""Stand-alone timings of the HBM-bound kernels of the hot path with their ALGORITHMIC bytes: each kernel alone on
the device, HIP events around back-to-back launches on the current stream.  bench.py reports these beside the in-step figures
(which are brackets on overlapped streams, i.e. they include contention with the convolutions); tools/hbm_bench.py prints them."""
import torch

from . import ops, synth


def warm(fn, seconds=0.3):
    """run ``fn`` back to back for ``seconds`` so that the measurement sees SUSTAINED clocks: an idle MI355X sits at ~100 MHz and takes tens
    of milliseconds of load to reach its working point (30 launches of a 0.9 ms kernel right after three warm-up calls measured 7 % slow:
    0.913 ms against 0.855 ms over 400 launches, which is also what the kernel takes inside the benchmark step)."""
    import time
    t0 = time.time()
    fn()
    torch.cuda.synchronize()
    while time.time() - t0 < seconds:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()


def _timeit(fn, n):
    warm(fn, 0.2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def hbm_kernels_standalone(D=64, H=120, W=160, n=20, device=None, peak_gbs=8000.0, only_attention=None):
    """{kernel: {"avg_launch_us", "algorithmic_mb_per_launch", "achieved_gbs", "frac"}} at one volume size (default cfg2/cfg3).
    ``only_attention=(2,)``: just warp_attention with these source counts (the cfg5-size entry of bench.py's joint line)."""
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    vox = D * H * W
    g = torch.Generator(device=dev).manual_seed(0)
    res = {}

    def report(name, sec, nbytes):
        res[name] = {"avg_launch_us": round(sec * 1e6, 1), "algorithmic_mb_per_launch": round(nbytes / 1e6, 2),
                     "achieved_gbs": round(nbytes / sec / 1e9, 1), "frac": round(nbytes / sec / 1e9 / peak_gbs, 4)}

    K = torch.from_numpy(synth.intrinsics(4 * H, 4 * W)).clone()
    K[:2] *= 0.25
    K = K.to(dev)
    poses = [torch.from_numpy(synth.camera_pose(v)).to(dev) for v in range(5)]
    dv = (torch.arange(D, dtype=torch.float32) * (9.9 / (D - 1)) + 0.1).to(dev)
    src = torch.randn(H, W, 32, device=dev, generator=g)
    ref = torch.randn(H, W, 32, device=dev, generator=g)
    kvs = [torch.randn(D, H, W, 32, device=dev, generator=g) for _ in range(4)]
    if only_attention:
        for ns in only_attention:
            mats = torch.stack([ops.cam_volume_mats(poses[j + 1], poses[0], K) for j in range(ns)])
            report("warp_attention N=%d" % ns, _timeit(lambda: ops.warp_attention(kvs[0], kvs[1:1 + ns], mats, dv, 0.1, 9.9 / (D - 1)), n),
                   4 * 16 * vox * (2 + 2 * ns))
        return res
    proj = ops.cam_sweep_proj(poses[1], poses[0], K)
    out = torch.empty(D, H, W, 32, device=dev)
    report("homo_warp_costvol", _timeit(lambda: ops.homo_warp_costvol(src, ref, proj, dv, D, out=out), n), 4 * (2 * 32 * H * W + 32 * vox))
    for ns in (1, 2, 3):
        mats = torch.stack([ops.cam_volume_mats(poses[j + 1], poses[0], K) for j in range(ns)])
        report("warp_attention N=%d" % ns, _timeit(lambda: ops.warp_attention(kvs[0], kvs[1:1 + ns], mats, dv, 0.1, 9.9 / (D - 1)), n),
               4 * 16 * vox * (2 + 2 * ns))                       # K_t, h, K_j, V_j
    xh, ru = kvs[0], kvs[1]
    st = torch.tensor([0.1, 1.1, -0.1, 0.9], device=dev)
    gm, bt = torch.ones(16, device=dev), torch.zeros(16, device=dev)
    report("gru_reset_apply", _timeit(lambda: ops.gru_reset_apply(xh, ru, st, gm, bt), n), 4 * vox * (32 + 16 + 32))
    o_raw = torch.randn(D, H, W, 16, device=dev, generator=g)
    report("gru_blend", _timeit(lambda: ops.gru_blend(xh, ru, o_raw, st, st, gm, bt, gm, bt, kvs[2], 32), n), 4 * vox * (16 + 16 + 16 + 16))
    lg = torch.randn(3, D, H, W, device=dev, generator=g)
    report("softargmin_up (T=3)", _timeit(lambda: ops.softargmin_up(lg, dv, 4), n), 4 * 3 * (vox + 2 * 16 * H * W))
    f = torch.randn(32, H, W, device=dev, generator=g)
    wm = torch.randn(32, 32, device=dev, generator=g)
    report("mix1x1", _timeit(lambda: ops.mix1x1(f, wm, None), n), 4 * 2 * 32 * H * W)
    return res
