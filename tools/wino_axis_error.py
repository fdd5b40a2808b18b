"""fp32 error of F(2,3) Winograd forms of the 3x3x3 convolution against float64: direct (K = 4 accumulation steps like the MFMA), two axes (depth, rows: csrc/conv3d_wino2.hip)
and three axes (F(2x2x2, 3x3x3), not built).  numpy emulation, 32 -> 32 channels, 8 x 8 x 16 voxels.   python tools/wino_axis_error.py"""
import numpy as np
rng = np.random.default_rng(0)
BT = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], np.float64)
G = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], np.float64)
AT = np.array([[1,1,1,0],[0,1,-1,-1]], np.float64)
C = 32; D=8; H=8; W=16
def run(seed, xs=1.0):
    r = np.random.default_rng(seed)
    x = (r.standard_normal((C, D+2, H+2, W+2))*xs).astype(np.float32)
    w = (r.standard_normal((C, C, 3,3,3))*0.05).astype(np.float32)
    # fp64 reference
    ref = np.zeros((C, D, H, W))
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                ref += np.einsum('oi,idhw->odhw', w[:,:,kd,kh,kw].astype(np.float64), x[:,kd:kd+D,kh:kh+H,kw:kw+W].astype(np.float64))
    # direct fp32 (sequential accumulation over taps then channels in fp32 via einsum float32)
    acc = np.zeros((C, D, H, W), np.float32)
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                for i0 in range(0, C, 4):      # K=4 steps like the MFMA
                    acc = acc + np.einsum('oi,idhw->odhw', w[:,i0:i0+4,kd,kh,kw], x[i0:i0+4,kd:kd+D,kh:kh+H,kw:kw+W]).astype(np.float32)
    e_dir = np.abs(acc - ref).max()
    f32 = np.float32
    def wino(axes):
        # axes: which of (d,h,w) are in Winograd form
        U = w.astype(np.float64)
        for ax, on in zip((2,3,4), axes):
            if on: U = np.moveaxis(np.tensordot(G, U, axes=(1, ax)), 0, ax)
        U = U.astype(f32)
        y = np.zeros((C, D, H, W), f32)
        sd = 2 if axes[0] else 1; sh = 2 if axes[1] else 1; sw = 2 if axes[2] else 1
        for d0 in range(0, D, sd):
            for h0 in range(0, H, sh):
                for w0 in range(0, W, sw):
                    patch = x[:, d0:d0+sd+2, h0:h0+sh+2, w0:w0+sw+2]
                    T = patch
                    for ax, on in zip((1,2,3), axes):
                        if on:      # fp32 adds
                            T = np.moveaxis(np.tensordot(BT.astype(f32), T, axes=(1, ax)).astype(f32), 0, ax)
                    # products: m[o, a,b,c] = sum over i and non-wino taps
                    nd = 4 if axes[0] else 3; nh = 4 if axes[1] else 3; nw = 4 if axes[2] else 3
                    if all(axes):
                        m = np.zeros((C,4,4,4), f32)
                        for i0 in range(0, C, 4):
                            m = m + np.einsum('oiabc,iabc->oabc', U[:,i0:i0+4], T[i0:i0+4]).astype(f32)
                        out = m
                    else:   # (d,h) wino, w direct
                        m = np.zeros((C,4,4,1), f32)
                        for kw in range(3):
                            for i0 in range(0, C, 4):
                                m = m + np.einsum('oiab,iabc->oabc', U[:,i0:i0+4,:,:,kw], T[i0:i0+4,:,:,kw:kw+1]).astype(f32)
                        out = m
                    for ax, on in zip((1,2,3), axes):
                        if on:
                            out = np.moveaxis(np.tensordot(AT.astype(f32), out, axes=(1, ax)).astype(f32), 0, ax)
                    y[:, d0:d0+sd, h0:h0+sh, w0:w0+sw] = out
        return np.abs(y - ref).max(), np.sqrt(((y-ref)**2).mean())
    e2 = wino((True, True, False)); e3 = wino((True, True, True))
    return e_dir, e2, e3, np.abs(ref).max()
for s in range(3):
    e_dir, e2, e3, mag = run(s)
    print("seed %d |ref| %.2f: direct %.3g  2-axis max %.3g rms %.3g  3-axis max %.3g rms %.3g  ratio max %.2f rms %.2f" % (s, mag, e_dir, e2[0], e2[1], e3[0], e3[1], e3[0]/e2[0], e3[1]/e2[1]))
