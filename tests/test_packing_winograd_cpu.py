"""CPU checks of the Winograd weight packers (estdepth_amd/packing.py): the packed buffers are unpacked with the lane / k-step / tap
indexing the kernels use (csrc/conv3d_wino.hip, conv3d_wino2.hip, conv2d_wino.hip) and the Winograd algebra the kernels implement
(input transform B^T x, product with the packed U, output transform A^T m) is evaluated in numpy on a small random map -- it must
reproduce the direct 3x3(x3) cross-correlation (networks/layers_op.py:10-39: Conv2d / Conv3d, padding 1).  No GPU involved: a packing
regression (tap order, K permutation, channel-half split, G matrix) shows up here before any kernel runs."""
import numpy as np
import pytest
import torch

from estdepth_amd import packing

BT = np.array([[1.0, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])     # input transform of F(2,3): t = B^T x
AT = np.array([[1.0, 1, 1, 0], [0, 1, -1, -1]])                                  # output transform: y = A^T m


def _unpack_k(packed_lane_major, cin=32):
    """[..., 2 quads, 64 lanes, 4] -> [..., 16 columns j, cin] using the kernels' K permutation ch(g, t), t = 4 q + e."""
    lead = packed_lane_major.shape[:-3]
    out = np.zeros(lead + (16, cin), packed_lane_major.dtype)
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        for t in range(8):
            out[..., j, packing._ch(32, g, t)] = packed_lane_major[..., t // 4, lane, t % 4]
    return out


def _direct3d(x, w):
    """x [Cin, D, H, W], w [Cout, Cin, 3, 3, 3] -> [Cout, D, H, W] (zero padding 1), float64"""
    cin, D, H, W = x.shape
    xp = np.zeros((cin, D + 2, H + 2, W + 2)); xp[:, 1:-1, 1:-1, 1:-1] = x
    y = np.zeros((w.shape[0], D, H, W))
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                y += np.einsum("oi,idhw->odhw", w[:, :, kd, kh, kw], xp[:, kd:kd + D, kh:kh + H, kw:kw + W])
    return y


def test_pack_conv3d_wino2x_reproduces_the_direct_convolution():
    """csrc/conv3d_wino2x.hip: v_mfma_f32_32x32x2_f32 with the weights as the A operand -- lane (k2, o) of half step (sd, kw, c, q), row-transform
    index sh, element e multiplies input channel 16 c + 8 q + 4 k2 + e of the voxel record into output channel o."""
    rng = np.random.default_rng(4)
    w = rng.standard_normal((32, 32, 3, 3, 3)) * 0.1
    main_idx, out_idx = list(rng.permutation(32)), list(rng.permutation(32))
    packed = packing.pack_conv3d_wino2x(torch.from_numpy(w).float(), main_idx, out_idx).numpy().reshape(4, 3, 2, 2, 4, 64, 4)   # [sd][kw][c][q][sh][lane][e]
    U = np.zeros((4, 3, 4, 32, 32), np.float32)               # [sd][kw][sh][o][record position]
    for lane in range(64):
        k2, o = lane >> 5, lane & 31
        for c in range(2):
            for q in range(2):
                for e in range(4):
                    U[:, :, :, o, 16 * c + 8 * q + 4 * k2 + e] = packed[:, :, c, q, :, lane, e]
    D, H, W = 4, 6, 5
    x = rng.standard_normal((32, D, H, W))                    # record position p of x = weight input channel main_idx[p]
    xp = np.zeros((32, D + 2, H + 2, W + 2)); xp[:, 1:-1, 1:-1, 1:-1] = x
    y = np.zeros((32, D, H, W))
    for d0 in range(0, D, 2):
        for h0 in range(0, H, 2):
            T = np.einsum("sd,th,cdhw->stcw", BT, BT, xp[:, d0:d0 + 4, h0:h0 + 4, :])
            m = np.zeros((4, 4, 32, W))
            for kw in range(3):
                m += np.einsum("stoc,stcw->stow", U[:, kw].astype(np.float64), T[:, :, :, kw:kw + W])
            y[:, d0:d0 + 2, h0:h0 + 2, :] = np.einsum("ds,et,stow->odew", AT, AT, m)     # MFMA row o = output position o
    ref = _direct3d(x, w[:, main_idx][out_idx])
    assert np.abs(y - ref).max() < 1e-5 * np.abs(ref).max()


def test_pack_conv3d_wino3_reproduces_the_direct_convolution():
    """csrc/conv3d_wino3.hip: F(2x2x2, 3x3x3).  Block ((4 sd + sh) * 2 + cc) * 2 + hh, half nh, tap pair sp: element f of lane (g, j) multiplies record
    position 16 cc + 4 g + 2 (hh ^ (g & 1)) + (f & 1) into output position 16 nh + j for the tap sw = 2 sp + (f >> 1)."""
    rng = np.random.default_rng(6)
    w = rng.standard_normal((32, 32, 3, 3, 3)) * 0.1
    main_idx, out_idx = list(rng.permutation(32)), list(rng.permutation(32))
    packed = packing.pack_conv3d_wino3(torch.from_numpy(w).float(), main_idx, out_idx).numpy().reshape(4, 4, 2, 2, 2, 2, 64, 4)   # [sd][sh][cc][hh][nh][sp][lane][f]
    U = np.zeros((4, 4, 4, 32, 32), np.float32)               # [sd][sh][sw][output position][record position]
    seen = np.zeros((32, 32), int)
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        for cc in range(2):
            for hh in range(2):
                for nh in range(2):
                    for sp in range(2):
                        for f in range(4):
                            pos = 16 * cc + 4 * g + 2 * (hh ^ (g & 1)) + (f & 1)
                            U[:, :, 2 * sp + (f >> 1), 16 * nh + j, pos] = packed[:, :, cc, hh, nh, sp, lane, f]
                            seen[16 * nh + j, pos] += 1
    assert (seen == 4).all()                                  # every (output, input) pair once per tap sw
    D, H, W = 4, 6, 6
    x = rng.standard_normal((32, D, H, W))
    xp = np.zeros((32, D + 2, H + 2, W + 2)); xp[:, 1:-1, 1:-1, 1:-1] = x
    y = np.zeros((32, D, H, W))
    for d0 in range(0, D, 2):
        for h0 in range(0, H, 2):
            for w0 in range(0, W, 2):
                T = np.einsum("sd,th,uw,cdhw->stuc", BT, BT, BT, xp[:, d0:d0 + 4, h0:h0 + 4, w0:w0 + 4])
                m = np.einsum("stuoc,stuc->stuo", U.astype(np.float64), T)
                y[:, d0:d0 + 2, h0:h0 + 2, w0:w0 + 2] = np.einsum("ds,et,fu,stuo->odef", AT, AT, AT, m)
    ref = _direct3d(x, w[:, main_idx][out_idx])
    assert np.abs(y - ref).max() < 1e-5 * np.abs(ref).max()


def test_pack_conv3d_wino3_extra_sums_the_depth_transforms_with_the_planes_output_coefficients():
    """[2 planes][4 sh][2 nh][64 lanes][4 sw]: lane (g, j) holds A^T[plane][g] * U[sd = g][sh][sw][out_idx[16 nh + j]]; summing lane groups (the k index of one MFMA)
    against the depth-transformed patch and finishing rows and columns with A^T reproduces the direct convolution of the scalar channel."""
    rng = np.random.default_rng(8)
    w = rng.standard_normal((32, 33, 3, 3, 3)) * 0.1
    out_idx = list(rng.permutation(32))
    px = packing.pack_conv3d_wino3_extra(torch.from_numpy(w).float(), 32, out_idx).numpy().astype(np.float64)
    x = rng.standard_normal((4, 4, 4))
    T = np.einsum("sd,th,uw,dhw->stu", BT, BT, BT, x)                # [sd][sh][sw]
    y = np.zeros((32, 2, 2, 2))
    for nh in range(2):
        for j in range(16):
            mpl = np.zeros((2, 4, 4))                                # [plane][sh][sw]: one MFMA each, k = lane group g = sd
            for g in range(4):
                mpl += px[:, :, nh, 16 * g + j, :] * T[g][None]
            y[16 * nh + j] = np.einsum("et,fu,ptu->pef", AT, AT, mpl)
    ref = np.zeros((32, 2, 2, 2))
    for d in range(2):
        for h in range(2):
            for ww in range(2):
                ref[:, d, h, ww] = np.einsum("odhw,dhw->o", w[out_idx][:, 32], x[d:d + 3, h:h + 3, ww:ww + 3])
    assert np.abs(y - ref).max() < 1e-6 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("n_out", [32, 16])
def test_pack_conv3d_wino2_reproduces_the_direct_convolution(n_out):
    rng = np.random.default_rng(3)
    w = rng.standard_normal((n_out, 32, 3, 3, 3)) * 0.1
    main_idx = list(rng.permutation(32))                      # the packers take index lists: exercise a non-trivial one
    out_idx = list(rng.permutation(n_out))
    packed = packing.pack_conv3d_wino2(torch.from_numpy(w).float(), main_idx, out_idx).numpy()     # [48][NH][2][64][4]
    nh = n_out // 16
    U = _unpack_k(packed.reshape(4, 3, 4, nh, 2, 64, 4))      # [sd][kw][sh][nh][j][ci]  (tap = (3 sd + kw) * 4 + sh)
    D, H, W = 4, 6, 5
    x = rng.standard_normal((32, D, H, W))                    # channel c of x = weight input channel main_idx[c]
    xp = np.zeros((32, D + 2, H + 2, W + 2)); xp[:, 1:-1, 1:-1, 1:-1] = x
    y = np.zeros((n_out, D, H, W))
    for d0 in range(0, D, 2):
        for h0 in range(0, H, 2):
            patch = xp[:, d0:d0 + 4, h0:h0 + 4, :]                                       # [ci, 4, 4, W + 2]
            T = np.einsum("sd,th,cdhw->stcw", BT, BT, patch)                             # [sd][sh][ci][W + 2]
            m = np.zeros((4, 4, nh, 16, W))
            for kw in range(3):
                m += np.einsum("sthjc,stcw->sthjw", U[:, kw].astype(np.float64), T[:, :, :, kw:kw + W])
            yy = np.einsum("ds,et,sthjw->dehjw", AT, AT, m)                              # [2][2][nh][j][W]
            y[:, d0:d0 + 2, h0:h0 + 2, :] = yy.reshape(2, 2, n_out, W).transpose(2, 0, 1, 3)  # output row 16 nh + j
    w_used = w[np.asarray(out_idx)][:, np.asarray(main_idx)]   # output row k = filter out_idx[k]; input slot c = channel main_idx[c]
    ref = _direct3d(x, w_used)
    assert np.abs(y - ref).max() < 2e-6 * max(1.0, np.abs(ref).max())


def test_pack_conv3d_wino_depth_axis_reproduces_the_direct_convolution():
    rng = np.random.default_rng(4)
    w = rng.standard_normal((32, 32, 3, 3, 3)) * 0.1
    idx = list(range(32))
    packed = packing.pack_conv3d_wino(torch.from_numpy(w).float(), idx, idx).numpy()               # [37][2][2][64][4]
    U = _unpack_k(packed[:36].reshape(4, 3, 3, 2, 2, 64, 4))   # [s][kh][kw][nh][j][ci]  (tap = 9 s + 3 kh + kw)
    assert not packed[36].any()                                # the padding tap
    D, H, W = 4, 3, 5
    x = rng.standard_normal((32, D, H, W))
    xp = np.zeros((32, D + 2, H + 2, W + 2)); xp[:, 1:-1, 1:-1, 1:-1] = x
    y = np.zeros((32, D, H, W))
    for d0 in range(0, D, 2):
        T = np.einsum("sd,cdhw->schw", BT, xp[:, d0:d0 + 4])                                  # [s][ci][H + 2][W + 2]
        m = np.zeros((4, 2, 16, H, W))
        for kh in range(3):
            for kw in range(3):
                m += np.einsum("snjc,schw->snjhw", U[:, kh, kw].astype(np.float64), T[:, :, kh:kh + H, kw:kw + W])
        yy = np.einsum("ds,snjhw->dnjhw", AT, m)
        y[:, d0:d0 + 2] = yy.reshape(2, 32, H, W).transpose(1, 0, 2, 3)
    assert np.abs(y - _direct3d(x, w)).max() < 2e-6 * max(1.0, np.abs(y).max())


@pytest.mark.parametrize("nt", [2, 4])
def test_pack_conv2d_wino_row_axis_reproduces_the_direct_convolution(nt):
    rng = np.random.default_rng(5)
    cin, cout = 64, 64
    w = rng.standard_normal((cout, cin, 3, 3)) * 0.1
    packed = packing.pack_conv2d_wino(torch.from_numpy(w).float(), nt).numpy()     # [groups][chunks][13][2 nt][64][4]
    groups, chunks = cout // (16 * nt), cin // 32
    assert not packed[:, :, 12].any()                          # the padding tap
    # quad index of (k-step t, N tile n) = (t nt + n) // 4, element (t nt + n) % 4; output channel = 16 nt grp + nt j + n
    U = np.zeros((4, 3, cout, cin))                            # [i][kw][co][ci]  (tap = 3 i + kw)
    for grp in range(groups):
        for c in range(chunks):
            for lane in range(64):
                g, j = lane >> 4, lane & 15
                for t in range(8):
                    for n in range(nt):
                        idx = t * nt + n
                        U[:, :, grp * 16 * nt + nt * j + n, c * 32 + packing._ch(32, g, t)] = \
                            packed[grp, c, :12, idx // 4, lane, idx % 4].reshape(4, 3)
    H, W = 6, 7
    x = rng.standard_normal((cin, H, W))
    xp = np.zeros((cin, H + 2, W + 2)); xp[:, 1:-1, 1:-1] = x
    y = np.zeros((cout, H, W))
    for h0 in range(0, H, 2):
        T = np.einsum("ih,chw->icw", BT, xp[:, h0:h0 + 4])     # [i][ci][W + 2]
        m = np.zeros((4, cout, W))
        for kw in range(3):
            m += np.einsum("ioc,icw->iow", U[:, kw], T[:, :, kw:kw + W])
        y[:, h0:h0 + 2] = np.einsum("ri,iow->orw", AT, m)
    ref = np.zeros((cout, H, W))
    for kh in range(3):
        for kw in range(3):
            ref += np.einsum("oi,ihw->ohw", w[:, :, kh, kw], xp[:, kh:kh + H, kw:kw + W])
    assert np.abs(y - ref).max() < 2e-6 * max(1.0, np.abs(ref).max())


def test_pack_conv3d_wino2_extra_scalar_channel():
    """33rd (scalar) input channel of the key|value convolution: [4 sd][2 halves][64 lanes][4 sh], lane group g = column tap kw (g = 3: zeros);
    evaluated like the kernel's extra stage it must add exactly the direct convolution of that channel."""
    rng = np.random.default_rng(6)
    w = rng.standard_normal((32, 33, 3, 3, 3)) * 0.1
    out_idx = list(range(32))
    packed = packing.pack_conv3d_wino2_extra(torch.from_numpy(w).float(), 32, out_idx).numpy()     # [4][2][64][4]
    U = np.zeros((4, 4, 32, 3))                               # [sd][sh][co][kw]
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        if g == 3:
            assert not packed[:, :, lane, :].any()
            continue
        for nh in range(2):
            U[:, :, 16 * nh + j, g] = packed[:, nh, lane, :]
    D, H, W = 4, 4, 5
    e = rng.standard_normal((1, D, H, W))
    ep = np.zeros((1, D + 2, H + 2, W + 2)); ep[:, 1:-1, 1:-1, 1:-1] = e
    y = np.zeros((32, D, H, W))
    for d0 in range(0, D, 2):
        for h0 in range(0, H, 2):
            T = np.einsum("sd,th,dhw->stw", BT, BT, ep[0, d0:d0 + 4, h0:h0 + 4, :])
            m = np.zeros((4, 4, 32, W))
            for kw in range(3):
                m += np.einsum("sto,stw->stow", U[..., kw], T[:, :, kw:kw + W])
            y[:, d0:d0 + 2, h0:h0 + 2, :] = np.einsum("ds,et,stow->odew", AT, AT, m)
    ref = _direct3d(e, w[:, 32:33])
    assert np.abs(y - ref).max() < 2e-6 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("k, cin, cout", [(1, 32, 48), (3, 16, 32)])
def test_pack_conv2d_small_and_to16_layouts(k, cin, cout):
    rng = np.random.default_rng(7)
    w = rng.standard_normal((cout, cin, k, k)).astype(np.float32)
    p = packing.pack_conv2d_small(torch.from_numpy(w)).numpy()                     # [cout/16][k*k][cin/16][64][4]
    for lane in (0, 17, 38, 63):
        g, i = lane >> 4, lane & 15
        for nt in range(cout // 16):
            for q in range(cin // 16):
                for ks in range(4):
                    assert np.array_equal(p[nt, :, q, lane, ks], w[16 * nt + i, 16 * q + 4 * g + ks].reshape(k * k))
    if k == 3:
        w16 = w[:16]
        p16 = packing.pack_conv2d_to16(torch.from_numpy(w16)).numpy()              # [9][cin/16][64][4]
        for lane in (3, 29, 50):
            g, i = lane >> 4, lane & 15
            for ks in range(4):
                assert np.array_equal(p16[:, 0, lane, ks], w16[i, 4 * g + ks].reshape(9))


def test_stem7x7_and_taps_packers_reproduce_the_convolution():
    """packing.pack_stem7x7 / pack_conv2d_taps read back with the indexing of csrc/conv2d_taps.hip (lane (g, i) at k-step s of window row ky
    multiplies slot 6 g + s = channel k % 3 of window pixel k // 3, pixel 7 a zero tap; taps (ky, kx) of [cout][cin] matrices) == conv2d."""
    import torch.nn.functional as F
    from estdepth_amd import packing
    g = torch.Generator().manual_seed(5)
    w = torch.randn(64, 3, 7, 7, generator=g)
    x = torch.randn(1, 3, 9, 9, generator=g)
    wp = packing.pack_stem7x7(w).numpy()                                  # [7][6][4][64]
    ref = F.conv2d(x.double(), w.double(), None, 2, 3)[0, :, 2, 2].numpy()        # output pixel (2, 2): window rows / columns 1..7
    win = np.zeros((7, 8, 3))
    win[:, :7] = x[0, :, 1:8, 1:8].permute(1, 2, 0).numpy()
    win[:, 7] = 1e6                                                       # whatever lies behind the window: must meet a zero weight
    out = np.zeros(64)
    for lane in range(64):
        gg, i = lane >> 4, lane & 15
        for ky in range(7):
            for s in range(6):
                k = 6 * gg + s
                for u in range(4):
                    out[16 * u + i] += float(wp[ky, s, u, lane]) * win[ky, k // 3, k % 3]
    assert np.abs(out - ref).max() < 1e-4
    w3 = torch.randn(32, 16, 3, 3, generator=g)
    wt = packing.pack_conv2d_taps(w3)
    assert tuple(wt.shape) == (9, 32, 16)
    for ky in range(3):
        for kx in range(3):
            assert torch.equal(wt[ky * 3 + kx], w3[:, :, ky, kx])


def test_pack_conv3d_xout_taps_reproduces_the_direct_convolution():
    """csrc/conv3d_xout.hip (round 6): ONE output channel of a 33-input-channel convolution with the 27 taps as the matrix core's rows -- lane (g, i)
    of (chunk c, tap tile t) holds w[out][main_idx[16 c + 4 g + e]][tap 16 t + i], the scalar channel's taps sit in lane group 0 of their own k-step.
    Unpacked with that indexing, P[u][tap] = sum_c w[c][tap] x[u][c] followed by the shifted sum y[v] = sum_tap P[v + offset(tap)][tap]
    (tap = (kd 3 + kh) 3 + kw) is the direct cross-correlation."""
    rng = np.random.default_rng(5)
    w = rng.standard_normal((33, 33, 3, 3, 3))
    main_idx, extra_idx, out_ch = list(range(1, 33)), 0, 32                 # dres2: scalar channel first in the weight, last output channel
    packed = packing.pack_conv3d_xout_taps(torch.from_numpy(w).float(), main_idx, extra_idx, out_ch).numpy().astype(np.float64)
    main, ext = packed[:1024].reshape(2, 2, 64, 4), packed[1024:].reshape(2, 64)
    wt = np.zeros((33, 32))                                                  # [channel in MEMORY order: 32 main then the scalar][tap row]
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        for t in range(2):
            for c in range(2):
                for e in range(4):
                    wt[16 * c + 4 * g + e, 16 * t + i] = main[c, t, lane, e]
            if g == 0:
                wt[32, 16 * t + i] = ext[t, lane]
            else:
                assert ext[t, lane] == 0.0                                   # the scalar channel's k-step is (s, 0, 0, 0)
    assert np.all(wt[:, 27:] == 0.0)
    D, H, W = 4, 5, 6
    x_main, x_s = rng.standard_normal((32, D, H, W)), rng.standard_normal((D, H, W))
    xm = np.concatenate([x_main, x_s[None]], 0)                              # memory order of the kernel's operands
    P = np.einsum("ct,cdhw->tdhw", wt, xm)                                   # the pointwise product
    Pp = np.zeros((32, D + 2, H + 2, W + 2)); Pp[:, 1:-1, 1:-1, 1:-1] = P
    y = np.zeros((D, H, W))
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                y += Pp[(kd * 3 + kh) * 3 + kw, kd:kd + D, kh:kh + H, kw:kw + W]
    x_ref = np.concatenate([x_s[None], x_main], 0)                           # the weight's channel order: scalar channel 0, then the 32 main ones
    ref = _direct3d(x_ref, w.astype(np.float32).astype(np.float64))[out_ch]          # (the packer stores float32)
    assert np.abs(y - ref).max() < 1e-5 * np.abs(ref).max()
