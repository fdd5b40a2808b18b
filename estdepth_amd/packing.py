"""Host-side packing of Conv3d weights into the MFMA fragment order consumed by
csrc/conv3d_mfma.hip, and BatchNorm folding.

Fragment order (v_mfma_f32_16x16x4_f32: lane l -> k group g = l >> 4, column j = l & 15):
  * input channel consumed by lane group g at k-step t of a tap:
        CM = 32:  ch(g,t) = 4g + t            for t < 4
                  ch(g,t) = 16 + 4g + (t-4)   for t >= 4
        CM = 16:  ch(g,t) = 4g + t
    (so a lane fetches its 8 A operands of a tap with two 16-byte LDS reads)
  * output channel position of (N tile n, column j):
        n_tiles = 1:  pos = j
        n_tiles == 2: pos = 2j + n   (n_tiles == 3 = these 32 channels + channel 32 on the VALU, pack_xout)
    (a lane owns two adjacent channels -> 8-byte epilogue stores that tile whole voxel records)
  * main buffer  : float32 [28 taps][QN][64 lanes][4]   QN = (CM/4)*n_tiles/4, flat index
                   idx = t*n_tiles + n -> (quad idx//4, element idx%4); tap 27 is zero padding that the
                   kernel's one-tap-ahead prefetch may read.
  * extra buffer : float32 [XQ][64 lanes][4] for the single scalar input channel whose 27 taps form
                   7 more k-steps: lane group g at step s multiplies tap 4s+g (tap 27 = zero).
"""
import numpy as np
import torch


def _ch(cm, g, t):
    if cm == 32:
        return 4 * g + t if t < 4 else 16 + 4 * g + (t - 4)
    return 4 * g + t


def _pos(n_tiles, n, j):
    if n_tiles == 1:
        return j
    if n < 2:
        return 2 * j + n
    return 32 if j == 0 else -1


def pack_xout(weight, main_idx, extra_idx, out_channel):
    """33rd output channel (n_tiles == 3): evaluated on the VALU from the A fragments, so its weights are packed in
    A order: [28 taps][2 quads][64 lanes][4] with element t of lane group g = w[out_channel][ch(g,t)][tap], followed by
    [2][64][4] = the extra input channel's taps 4s+g (s = 0..6)."""
    w = weight.detach().float().cpu().numpy().reshape(weight.shape[0], weight.shape[1], 27)
    out = np.zeros((28 * 2 + 2, 64, 4), np.float32)
    for lane in range(64):
        g = lane >> 4
        for t in range(8):
            out[np.arange(27) * 2 + t // 4, lane, t % 4] = w[out_channel, main_idx[_ch(32, g, t)], :]
        if extra_idx is not None:
            for s_ in range(7):
                tap = 4 * s_ + g
                if tap <= 26:
                    out[56 + s_ // 4, lane, s_ % 4] = w[out_channel, extra_idx, tap]
    return torch.from_numpy(out)


def pack_conv3d_xout_taps(weight, main_idx, extra_idx, out_channel):
    """ONE output channel of a 33-input-channel 3x3x3 convolution for csrc/conv3d_xout.hip (the taps are the matrix core's rows): float32
    [2 chunks][2 tap tiles][64 lanes][4] -- lane (g, i) of (chunk c, tile t) holds w[out_channel][main_idx[16 c + 4 g + e]][tap 16 t + i], e = 0..3
    (taps 27..31 are zero) -- followed by [2 tap tiles][64 lanes]: the scalar input channel's weight of tap 16 t + i for lane group g = 0, zero
    for the others (its k-step is (s, 0, 0, 0)).  Tap = (kd * 3 + kh) * 3 + kw."""
    w = weight.detach().double().cpu().numpy().reshape(weight.shape[0], weight.shape[1], 27)
    out = np.zeros(4 * 64 * 4 + 2 * 64, np.float32)
    main = out[:1024].reshape(2, 2, 64, 4)
    ext = out[1024:].reshape(2, 64)
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        for t in range(2):
            tap = 16 * t + i
            if tap > 26:
                continue
            for c in range(2):
                for e in range(4):
                    main[c, t, lane, e] = w[out_channel, main_idx[16 * c + 4 * g + e], tap]
            if g == 0 and extra_idx is not None:
                ext[t, lane] = w[out_channel, extra_idx, tap]
    return torch.from_numpy(out)


def pack_conv3d(weight, main_idx, extra_idx, out_idx, n_tiles):
    """weight: [Cout, Cin, 3,3,3] tensor.  main_idx: list of 16/32 input-channel indices (order = main
    channel order in memory).  extra_idx: index of the scalar input channel or None.
    out_idx: original output channel for each output position.  Returns (w_main, w_extra|None) float32 CPU.
    n_tiles == 3 (32 + one channel) packs the MFMA part with 2 tiles; the 33rd channel goes through pack_xout."""
    if n_tiles == 3:
        n_tiles = 2
        out_idx = list(out_idx)[:32]
    w = weight.detach().float().cpu().numpy().reshape(weight.shape[0], weight.shape[1], 27)
    cm = len(main_idx)
    assert cm in (16, 32)
    ks = cm // 4
    qn = ks * n_tiles // 4
    assert ks * n_tiles % 4 == 0
    main = np.zeros((28, qn, 64, 4), np.float32)
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        for t in range(ks):
            ci = main_idx[_ch(cm, g, t)]
            for n in range(n_tiles):
                pos = _pos(n_tiles, n, j)
                if pos < 0 or pos >= len(out_idx):
                    continue
                idx = t * n_tiles + n
                main[:27, idx // 4, lane, idx % 4] = w[out_idx[pos], ci, :]
    extra = None
    if extra_idx is not None:
        xq = (7 * n_tiles + 3) // 4
        extra = np.zeros((xq, 64, 4), np.float32)
        for lane in range(64):
            g, j = lane >> 4, lane & 15
            for s in range(7):
                tap = 4 * s + g
                if tap > 26:
                    continue
                for n in range(n_tiles):
                    pos = _pos(n_tiles, n, j)
                    if pos < 0 or pos >= len(out_idx):
                        continue
                    idx = s * n_tiles + n
                    extra[idx // 4, lane, idx % 4] = w[out_idx[pos], extra_idx, tap]
        extra = torch.from_numpy(extra)
    return torch.from_numpy(main), extra


def fold_bn_fp32(bn, out_idx):
    """Same folding done in fp32 exactly like ATen's native_batch_norm (invstd = 1/sqrt(var+eps))."""
    g = bn.weight.detach().float().cpu()
    b = bn.bias.detach().float().cpu()
    m = bn.running_mean.detach().float().cpu()
    v = bn.running_var.detach().float().cpu()
    invstd = 1.0 / torch.sqrt(v + bn.eps)
    sc = g * invstd
    sh = b - m * sc
    idx = torch.as_tensor(out_idx, dtype=torch.long)
    return sc[idx].contiguous(), sh[idx].contiguous()


def pack_conv2d(weight, group_tiles):
    """3x3 Conv2d weight [Cout, Cin, 3, 3] (Cin, Cout multiples of 32) -> float32
    [Cout/(16*NT)][Cin/32][10 taps (9 + zero pad)][2*NT quads][64 lanes][4] for csrc/conv2d_mfma.hip:
    lane (g, j), k-step t, N tile n  ->  quad (t*NT+n)//4, element (t*NT+n)%4,
    input channel 32*chunk + ch(g,t) (same K permutation as the 3D kernel), output channel 16*NT*grp + NT*j + n."""
    nt = group_tiles
    w = weight.detach().float().cpu().numpy()
    cout, cin = w.shape[:2]
    assert cin % 32 == 0 and cout % (16 * nt) == 0 and w.shape[2:] == (3, 3)
    w = w.reshape(cout, cin, 9)
    groups, chunks = cout // (16 * nt), cin // 32
    out = np.zeros((groups, chunks, 10, 2 * nt, 64, 4), np.float32)
    lane = np.arange(64)
    g, j = lane >> 4, lane & 15
    for t in range(8):
        ci_in_chunk = np.array([_ch(32, int(gg), t) for gg in g])           # [64]
        for n in range(nt):
            idx = t * nt + n
            for grp in range(groups):
                co = grp * 16 * nt + nt * j + n                             # [64]
                for c in range(chunks):
                    # value[tap, lane] = w[co[lane], c*32 + ci[lane], tap]
                    out[grp, c, :9, idx // 4, :, idx % 4] = w[co, c * 32 + ci_in_chunk, :].T
    return torch.from_numpy(out)


def bf16_split3(x):
    """float32 array -> three uint16 arrays (bf16 bit patterns, round-to-nearest-even) with x == p0 + p1 + p2 exactly
    for every finite x whose pieces stay normal (3 x 8 significand bits cover fp32's 24)."""
    def rne(v):
        u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)
    r = np.ascontiguousarray(x, dtype=np.float32)
    pieces = []
    for _ in range(3):
        p = rne(r)
        pieces.append((p.view(np.uint32) >> 16).astype(np.uint16))
        r = (r - p).astype(np.float32)
    return pieces


def pack_conv3d_split(weight, main_idx, out_idx, extra_idx=None, n_tiles=2):
    """Split weights for csrc/conv3d_split_bf16.hip: int16 [27|28 records][4096] (bf16 bit patterns), 8192 bytes per tap:
      bytes    0..6143  [3 pieces][NB n-tiles][64 lanes][8] (NB = 1 for n_tiles == 1, else 2): v_mfma_f32_16x16x32_bf16 B
                        operand, lane l = column j = l & 15, k = 8*(l >> 4) .. +7 = position of the input channel in memory
                        order (main_idx); output channel position of (tile n, column j) = NB*j + n;
      bytes 6144..6335  [3 pieces][4 k-groups][8]: the same k order for output channel out_idx[32] (n_tiles == 3), else 0;
      rest              zero (the kernel reads a zero block at byte 6400).
    With ``extra_idx`` a 28th record holds the scalar input channel: k = tap index 0..26 (27..31 zero)."""
    w = weight.detach().float().cpu().numpy().reshape(weight.shape[0], weight.shape[1], 27)
    assert len(main_idx) == 32 and n_tiles in (1, 2, 3) and len(out_idx) >= (16 if n_tiles == 1 else 32)
    nb = 1 if n_tiles == 1 else 2
    ntaps = 27 if extra_idx is None else 28
    main = np.zeros((ntaps, nb, 64, 8), np.float32)
    xcol = np.zeros((ntaps, 4, 8), np.float32)
    for lane in range(64):
        kg, j = lane >> 4, lane & 15
        for n in range(nb):
            co = out_idx[nb * j + n]
            for e in range(8):
                main[:27, n, lane, e] = w[co, main_idx[8 * kg + e], :]
                if extra_idx is not None and 8 * kg + e < 27:
                    main[27, n, lane, e] = w[co, extra_idx, 8 * kg + e]
    if n_tiles == 3:
        co = out_idx[32]
        for kg in range(4):
            for e in range(8):
                xcol[:27, kg, e] = w[co, main_idx[8 * kg + e], :]
                if extra_idx is not None and 8 * kg + e < 27:
                    xcol[27, kg, e] = w[co, extra_idx, 8 * kg + e]
    rec = np.zeros((ntaps, 4096), np.uint16)
    rec[:, :1536 * nb] = np.stack(bf16_split3(main), axis=1).reshape(ntaps, 1536 * nb)
    rec[:, 3072:3072 + 96] = np.stack(bf16_split3(xcol), axis=1).reshape(ntaps, 96)
    return torch.from_numpy(rec.view(np.int16).copy())


def pack_conv2d_wino2(weight):
    """3x3 Conv2d weight [Cout, Cin, 3, 3] for csrc/conv2d_wino2.hip: both axes in Winograd F(2,3) form, U = G g G^T with
    G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] on kh and on kw (float64, rounded once to float32), packed as float32
    [Cout/32][Cin/32][8 steps = 2 sh + channel half][4 sw][2 output tiles][64 lanes][4]: element e of lane (g, j) of (group, chunk, step,
    sw, tile nt) = U[sh][sw][32 group + 16 nt + j][32 chunk + 16 half + 4 g + e] -- output channel j as the MFMA's M row, the four
    consecutive input channels of the lane's 16-byte chunk as its four k-steps."""
    w = weight.detach().double().cpu().numpy()
    cout, cin = w.shape[:2]
    assert cin % 32 == 0 and cout % 32 == 0 and w.shape[2:] == (3, 3)
    G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])
    U = np.einsum("sh,tw,oihw->stoi", G, G, w).astype(np.float32)           # [4 sh, 4 sw, Cout, Cin]
    groups, chunks = cout // 32, cin // 32
    out = np.zeros((groups, chunks, 8, 4, 2, 64, 4), np.float32)
    lane = np.arange(64)
    g, j = lane >> 4, lane & 15
    for grp in range(groups):
        for c in range(chunks):
            for st in range(8):
                sh, half = st >> 1, st & 1
                for nt in range(2):
                    co = grp * 32 + nt * 16 + j                              # [64]
                    for e in range(4):
                        ci = c * 32 + 16 * half + 4 * g + e                  # [64]
                        out[grp, c, st, :, nt, :, e] = U[sh][:, co, ci]
    return torch.from_numpy(out)


def pack_conv2d_split(weight):
    """3x3 Conv2d weight [Cout, Cin, 3, 3] (multiples of 32) for csrc/conv2d_split_bf16.hip: int16
    [Cout/32 groups][Cin/32 chunks][9 taps][4096]: per record bytes 0..6143 = [3 pieces][2 n-tiles][64 lanes][8] bf16
    (lane l = column j = l & 15, k = 8*(l >> 4) .. +7 inside the chunk; output channel = 32*group + 2j + n-tile), rest zero."""
    w = weight.detach().float().cpu().numpy()
    cout, cin = w.shape[:2]
    assert cout % 32 == 0 and cin % 32 == 0 and w.shape[2:] == (3, 3)
    w = w.reshape(cout // 32, 16, 2, cin // 32, 4, 8, 9)            # [grp][j][n][chunk][kg][e][tap]
    sel = np.ascontiguousarray(w.transpose(0, 3, 6, 2, 4, 1, 5))    # [grp][chunk][tap][n][kg][j][e]
    sel = sel.reshape(cout // 32, cin // 32, 9, 2, 64, 8)           # lane = kg*16 + j
    rec = np.zeros((cout // 32, cin // 32, 9, 4096), np.uint16)
    rec[..., :3072] = np.stack(bf16_split3(sel), axis=3).reshape(cout // 32, cin // 32, 9, 3072)
    return torch.from_numpy(rec.view(np.int16).copy())


def pack_conv3d_wino_extra(weight, extra_idx, out_idx):
    """the scalar 33rd input channel of a 33 -> 32 convolution for csrc/conv3d_wino.hip<EXTRA>: float32
    [2 channel halves][3 quads][64 lanes][4]; element 3 s + k of lane (g, j) of half nh = U_s[out_idx[16 nh + j]][extra_idx] at
    tap 4 k + g of the 3x3 (kh, kw) window (taps 9..11: zero), U_s as in pack_conv3d_wino."""
    w = weight.detach().double().cpu().numpy()[:, extra_idx]         # [Cout, kd, kh, kw]
    g0, g1, g2 = w[:, 0], w[:, 1], w[:, 2]
    U = np.stack([g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2], 0).astype(np.float32).reshape(4, w.shape[0], 9)
    out = np.zeros((2, 3, 64, 4), np.float32)
    oi = np.asarray(out_idx)
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        for nh in range(2):
            for s_ in range(4):
                for k in range(3):
                    tap = 4 * k + g
                    if tap < 9:
                        idx = 3 * s_ + k
                        out[nh, idx // 4, lane, idx % 4] = U[s_, oi[16 * nh + j], tap]
    return torch.from_numpy(out)


def pack_conv3d_wino_xout(weight, main_idx, extra_idx, out_ch):
    """the 33rd OUTPUT channel of a 33 -> 33 convolution for csrc/conv3d_wino.hip<EXTRA, XOUT>: float32
    [36 taps (s, kh, kw)][2 quads][4 lane groups][4] -- element r of (tap, q, g) = U_s[out_ch][main_idx[16 q + 4 g + r]] at (kh, kw),
    the channel an A-fragment lane of group g holds there -- followed by the scalar input channel's [3 quads][4 lane groups][4]
    (element 3 s + k of group g = U_s[out_ch][extra_idx] at tap 4 k + g; taps 9..11: zero)."""
    w = weight.detach().double().cpu().numpy()[out_ch]               # [Cin, kd, kh, kw]
    g0, g1, g2 = w[:, 0], w[:, 1], w[:, 2]
    U = np.stack([g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2], 0).astype(np.float32)    # [4, Cin, 3, 3]
    mi = np.asarray(main_idx)
    out = np.zeros((36 * 2 + 3, 4, 4), np.float32)
    for s_ in range(4):
        for t in range(9):
            for q in range(2):
                for g in range(4):
                    out[(s_ * 9 + t) * 2 + q, g] = U[s_, mi[16 * q + 4 * g:16 * q + 4 * g + 4], t // 3, t % 3]
    Ux = U[:, extra_idx].reshape(4, 9)
    for g in range(4):
        for s_ in range(4):
            for k in range(3):
                tap = 4 * k + g
                if tap < 9:
                    idx = 3 * s_ + k
                    out[72 + idx // 4, g, idx % 4] = Ux[s_, tap]
    return torch.from_numpy(out)


def pack_conv3d_wino(weight, main_idx, out_idx):
    """32 -> 32 filters for csrc/conv3d_wino.hip: the depth taps g0, g1, g2 of every (kh, kw) column in Winograd F(2,3) form
    U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2 (evaluated in float64, rounded once to float32), packed
    as float32 [37 taps][2 channel halves][2 quads][64 lanes][4]: tap = 9 s + 3 kh + kw (s = transform index), lane (g, j) of
    half nh holds, at k-step t = 4 q + e, U_s[out_idx[16 nh + j]][main_idx[ch(g, t)]][kh][kw]; tap 36 is zero padding that
    the kernel's one-tap-ahead prefetch may read."""
    assert len(main_idx) == 32 and len(out_idx) == 32
    w = weight.detach().double().cpu().numpy()                       # [Cout, Cin, kd, kh, kw]
    g0, g1, g2 = w[:, :, 0], w[:, :, 1], w[:, :, 2]
    U = np.stack([g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2], 0).astype(np.float32)       # [4, Cout, Cin, 3, 3]
    U = U.reshape(4, U.shape[1], U.shape[2], 9)
    out = np.zeros((37, 2, 2, 64, 4), np.float32)
    oi, mi = np.asarray(out_idx), np.asarray(main_idx)
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        for t in range(8):
            ci = mi[_ch(32, g, t)]
            for nh in range(2):
                # [4 s, 9 taps] -> taps 9 s + k
                out[:36, nh, t // 4, lane, t % 4] = U[:, oi[16 * nh + j], ci, :].reshape(36)
    return torch.from_numpy(out)


def pack_conv3d_wino2(weight, main_idx, out_idx):
    """32 -> 32 (or 32 -> 16) filters for csrc/conv3d_wino2.hip: depth AND row taps in Winograd F(2,3) form, U = G g G^T with
    G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] applied on kd and on kh (float64, rounded once to float32), packed as float32
    [48 taps][NH channel halves][2 quads][64 lanes][4] (NH = len(out_idx) / 16 = 2 | 1): tap = (3 sd + kw) * 4 + sh (sd / sh = depth /
    row transform index); lane (g, j) of half nh holds, at k-step t = 4 q + e, U[sd][sh][out_idx[16 nh + j]][main_idx[ch(g, t)]][kw]."""
    assert len(main_idx) == 32 and len(out_idx) in (16, 32)
    nhalf = len(out_idx) // 16
    w = weight.detach().double().cpu().numpy()                       # [Cout, Cin, kd, kh, kw]
    G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])
    U = np.einsum("sd,th,oidhw->stoiw", G, G, w).astype(np.float32)  # [4 sd, 4 sh, Cout, Cin, 3 kw]
    out = np.zeros((4, 3, 4, nhalf, 2, 64, 4), np.float32)           # [sd][kw][sh][nh][q][lane][e]
    oi, mi = np.asarray(out_idx), np.asarray(main_idx)
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        for t in range(8):
            ci = mi[_ch(32, g, t)]
            for nh in range(nhalf):
                out[:, :, :, nh, t // 4, lane, t % 4] = U[:, :, oi[16 * nh + j], ci, :].transpose(0, 2, 1)
    return torch.from_numpy(out.reshape(48, nhalf, 2, 64, 4))


def pack_conv3d_wino2x(weight, main_idx, out_idx):
    """32 -> 32 filters for csrc/conv3d_wino2x.hip (v_mfma_f32_32x32x2_f32, weights as the A operand): U = G g G^T over (kd, kh) as in
    pack_conv3d_wino2, packed as float32 [4 sd][3 kw][2 chunks c][2 q][4 sh][64 lanes][4]: element e of lane (k2 = lane >> 5, o = lane & 31)
    = U[sd][sh][out_idx[o]][main_idx[16 c + 8 q + 4 k2 + e]][kw] -- MFMA row o = output channel o, the lane's four consecutive input
    channels = the four k-steps that consume one 16-byte piece of a voxel record."""
    assert len(main_idx) == 32 and len(out_idx) == 32
    w = weight.detach().double().cpu().numpy()                       # [Cout, Cin, kd, kh, kw]
    G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])
    U = np.einsum("sd,th,oidhw->stoiw", G, G, w).astype(np.float32)  # [4 sd, 4 sh, Cout, Cin, 3 kw]
    oi, mi = np.asarray(out_idx), np.asarray(main_idx)
    out = np.zeros((4, 3, 2, 2, 4, 64, 4), np.float32)               # [sd][kw][c][q][sh][lane][e]
    for lane in range(64):
        k2, o = lane >> 5, lane & 31
        for c in range(2):
            for q in range(2):
                for e in range(4):
                    out[:, :, c, q, :, lane, e] = U[:, :, oi[o], mi[16 * c + 8 * q + 4 * k2 + e], :].transpose(0, 2, 1)
    return torch.from_numpy(out.reshape(4 * 3 * 2 * 2, 4, 64, 4))


def pack_conv3d_wino3(weight, main_idx, out_idx):
    """32 -> 32 filters for csrc/conv3d_wino3.hip: ALL THREE axes in Winograd F(2,3) form, U = G g G^T applied on kd, kh and kw (float64, rounded
    once to float32): 64 products (sd, sh, sw) of [32 out][32 in].  Packed as float32 [block = ((4 sd + sh) * 2 + cc) * 2 + hh][2 halves nh][2 tap
    pairs sp][64 lanes][4]: element f of lane (g, j) = U[sd][sh][sw = 2 sp + (f >> 1)][out_idx[16 nh + j]][main_idx[16 cc + 4 g + 2 (hh ^ (g & 1)) + (f & 1)]] --
    a block is what one half-sub-step of 8 MFMAs (4 taps sw x 2 k-steps) reads; lane groups g, g + 1 take the two channel pairs of their 16-byte
    chunk in opposite order (their 8-byte fragment reads then fall into different LDS banks)."""
    assert len(main_idx) == 32 and len(out_idx) == 32
    w = weight.detach().double().cpu().numpy()                       # [Cout, Cin, kd, kh, kw]
    G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])
    U = np.einsum("sd,th,uw,oidhw->stuoi", G, G, G, w).astype(np.float32)      # [4 sd, 4 sh, 4 sw, Cout, Cin]
    oi, mi = np.asarray(out_idx), np.asarray(main_idx)
    out = np.zeros((4, 4, 2, 2, 2, 2, 64, 4), np.float32)            # [sd][sh][cc][hh][nh][sp][lane][f]
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        for cc in range(2):
            for hh in range(2):
                for nh in range(2):
                    for sp in range(2):
                        for f in range(4):
                            ci = mi[16 * cc + 4 * g + 2 * (hh ^ (g & 1)) + (f & 1)]
                            out[:, :, cc, hh, nh, sp, lane, f] = U[:, :, 2 * sp + (f >> 1), oi[16 * nh + j], ci]
    return torch.from_numpy(out.reshape(64, 2, 2, 64, 4))


def pack_conv3d_wino3_extra(weight, extra_idx, out_idx):
    """the scalar 33rd input channel of a 33 -> 32 convolution for csrc/conv3d_wino3.hip<EXTRA>: float32 [2 output planes][4 sh][2 halves nh][64 lanes][4 sw].
    Element sw of lane (g, j) = A^T[plane][g] * U[sd = g][sh][sw][out_idx[16 nh + j]][extra_idx] with U = G g G^T on kd, kh and kw as in pack_conv3d_wino3 and
    A^T = [[1, 1, 1, 0], [0, 1, -1, -1]] the depth axis' output transform: the kernel sums the four depth transforms of the scalar channel inside ONE MFMA per
    (plane, sh, sw) -- the k index of the MFMA is the depth-transform index."""
    assert len(out_idx) == 32
    w = weight.detach().double().cpu().numpy()[:, extra_idx]         # [Cout, kd, kh, kw]
    G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])
    AT = np.array([[1.0, 1.0, 1.0, 0.0], [0.0, 1.0, -1.0, -1.0]])
    U = np.einsum("sd,th,uw,odhw->stuo", G, G, G, w)                 # [4 sd, 4 sh, 4 sw, Cout]
    oi = np.asarray(out_idx)
    out = np.zeros((2, 4, 2, 64, 4), np.float32)
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        for nh in range(2):
            for pl in range(2):
                out[pl, :, nh, lane, :] = (AT[pl, g] * U[g, :, :, oi[16 * nh + j]]).astype(np.float32)
    return torch.from_numpy(out)


def pack_conv3d_wino2_c16(weight, main_idx, out_idx):
    """16 -> 16 filters (the stereo heads) for csrc/conv3d_wino2_c16.hip: U = G g G^T over (kd, kh) as in pack_conv3d_wino2, packed as
    float32 [48 taps = (3 sd + kw) * 4 + sh][64 lanes][4]: element e of lane (g, j) = U[sd][sh][out_idx[j]][main_idx[4 g + e]][kw] -- output
    channel j as the MFMA's M row, the four consecutive input channels of the lane's 16-byte chunk as its four k-steps."""
    assert len(main_idx) == 16 and len(out_idx) == 16
    w = weight.detach().double().cpu().numpy()                       # [Cout, Cin, kd, kh, kw]
    G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])
    U = np.einsum("sd,th,oidhw->stoiw", G, G, w).astype(np.float32)  # [4 sd, 4 sh, Cout, Cin, 3 kw]
    out = np.zeros((4, 3, 4, 64, 4), np.float32)                     # [sd][kw][sh][lane][e]
    oi, mi = np.asarray(out_idx), np.asarray(main_idx)
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        for e in range(4):
            out[:, :, :, lane, e] = U[:, :, oi[j], mi[4 * g + e], :].transpose(0, 2, 1)
    return torch.from_numpy(out.reshape(48, 64, 4))


def pack_conv3d_wino2_extra(weight, extra_idx, out_idx):
    """the scalar 33rd input channel of a 33 -> 32 convolution for csrc/conv3d_wino2.hip<EXTRA>: float32
    [4 sd][2 channel halves][64 lanes][4 sh]; element sh of lane (g, j) of (sd, half nh) = U[sd][sh][out_idx[16 nh + j]][extra_idx]
    at column tap kw = g (g = 3: zero), U = G g G^T over (kd, kh) as in pack_conv3d_wino2."""
    w = weight.detach().double().cpu().numpy()[:, extra_idx]         # [Cout, kd, kh, kw]
    G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])
    U = np.einsum("sd,th,odhw->stow", G, G, w).astype(np.float32)    # [4 sd, 4 sh, Cout, 3 kw]
    out = np.zeros((4, 2, 64, 4), np.float32)
    oi = np.asarray(out_idx)
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        if g < 3:
            for nh in range(2):
                out[:, nh, lane, :] = U[:, :, oi[16 * nh + j], g]
    return torch.from_numpy(out)


def pack_conv3d_wino2_xout(weight, main_idx, extra_idx, out_ch):
    """the 33rd OUTPUT channel of a 33 -> 33 convolution for csrc/conv3d_wino2.hip<EXTRA, XOUT>: float32
    [24 steps = (3 sd + kw) * 2 + c][4 sh][4 lane groups][4] -- element j of (step, sh, g) = U[sd][sh][out_ch][main_idx[16 c + 4 g + j]][kw], the
    channels a B-fragment lane of group g holds in chunk c -- followed by the scalar input channel's [4 sd][4 lane groups][4 sh]
    (element sh of (sd, g) = U[sd][sh][out_ch][extra_idx] at column tap kw = g; g = 3: zero).  U = G g G^T over (kd, kh) as in
    pack_conv3d_wino2."""
    w = weight.detach().double().cpu().numpy()[out_ch]               # [Cin, kd, kh, kw]
    G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])
    U = np.einsum("sd,th,idhw->stiw", G, G, w).astype(np.float32)    # [4 sd, 4 sh, Cin, 3 kw]
    mi = np.asarray(main_idx)
    out = np.zeros((24 * 4 + 4, 4, 4), np.float32)
    for sd in range(4):
        for kw in range(3):
            for c in range(2):
                step = (3 * sd + kw) * 2 + c
                for sh in range(4):
                    for g in range(4):
                        out[step * 4 + sh, g] = U[sd, sh, mi[16 * c + 4 * g:16 * c + 4 * g + 4], kw]
        for g in range(3):
            out[96 + sd, g] = U[sd, :, extra_idx, g]
    return torch.from_numpy(out)


def pack_conv2d_to16(weight):
    """Conv2d weight [16, cin, 3, 3] (cin = 16 | 32) for csrc/refine2d.hip conv2d_k3_to16_kernel: float32 [9 taps][cin/16][64 lanes][4];
    element ks of lane (g, i) of (tap, half q) = weight[i][16 q + 4 g + ks][ky][kx] -- output channel i as the MFMA's M row, the
    four consecutive input channels a lane loads as its four k-steps."""
    w = weight.detach().float().cpu().numpy()
    cout, cin = w.shape[:2]
    assert cout == 16 and cin in (16, 32) and w.shape[2:] == (3, 3)
    out = np.zeros((9, cin // 16, 64, 4), np.float32)
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        for q in range(cin // 16):
            for ks in range(4):
                out[:, q, lane, ks] = w[i, 16 * q + 4 * g + ks].reshape(9)
    return torch.from_numpy(out)


def pack_conv2d_small(weight):
    """Conv2d weight [cout, cin, k, k] (k = 1 | 3; cin, cout multiples of 16) for csrc/refine2d.hip conv2d_small_kernel: float32
    [cout/16 tiles][k*k taps][cin/16][64 lanes][4]; element ks of lane (g, i) of (tile nt, tap, group q) =
    weight[16 nt + i][16 q + 4 g + ks][ky][kx] -- output channel i of the tile as the MFMA's M row, the four consecutive input
    channels a lane loads as its four k-steps."""
    w = weight.detach().float().cpu().numpy()
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    assert cout % 16 == 0 and cin % 16 == 0 and k in (1, 3) and w.shape[3] == k
    out = np.zeros((cout // 16, k * k, cin // 16, 64, 4), np.float32)
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        for nt in range(cout // 16):
            for q in range(cin // 16):
                for ks in range(4):
                    out[nt, :, q, lane, ks] = w[16 * nt + i, 16 * q + 4 * g + ks].reshape(k * k)
    return torch.from_numpy(out)


def pack_conv2d_taps(weight):
    """Conv2d weight [cout, cin, k, k] for csrc/conv2d_taps.hip: float32 [k*k taps (ky, kx)][cout][cin] -- per tap the [cout][cin]
    matrix the 1x1 kernel reads as it lies."""
    w = weight.detach().float()
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    assert w.shape[3] == k
    return w.permute(2, 3, 0, 1).reshape(k * k, cout, cin).contiguous().cpu()


def pack_stem7x7(weight):
    """Conv2d(3, 64, 7) weight [64, 3, 7, 7] for csrc/conv2d_taps.hip stem7x7s2_nhwc_kernel: float32 [7 rows][6 k-steps][4 channel
    tiles][64 lanes]; lane (g, i) at k-step s of row ky multiplies slot k = 6 g + s of the window row = channel k % 3 of window
    pixel kx = k // 3 (kx = 7, the second pixel of lane group 3, is a zero tap) into output channel 16 u + i."""
    w = weight.detach().float().cpu().numpy()
    assert w.shape == (64, 3, 7, 7)
    out = np.zeros((7, 6, 4, 64), np.float32)
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        for s in range(6):
            k = 6 * g + s
            kx, ch = k // 3, k % 3
            if kx < 7:
                for u in range(4):
                    out[:, s, u, lane] = w[16 * u + i, ch, :, kx]
    return torch.from_numpy(out)


def pack_conv2d_wino(weight, group_tiles):
    """3x3 Conv2d weight [Cout, Cin, 3, 3] for csrc/conv2d_wino.hip: the row taps g0, g1, g2 of every kw column in Winograd
    F(2,3) form U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2 (float64, rounded once to float32), packed as
    float32 [Cout/(16*NT)][Cin/32][13 taps (12 + zero pad)][2*NT quads][64 lanes][4] with tap = 3 i + kw and the lane / k-step /
    N-tile indexing of pack_conv2d."""
    nt = group_tiles
    w = weight.detach().double().cpu().numpy()
    cout, cin = w.shape[:2]
    assert cin % 32 == 0 and cout % (16 * nt) == 0 and w.shape[2:] == (3, 3)
    g0, g1, g2 = w[:, :, 0], w[:, :, 1], w[:, :, 2]                                        # [Cout, Cin, kw]
    U = np.stack([g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2], 2).astype(np.float32)   # [Cout, Cin, 4, 3]
    U = U.reshape(cout, cin, 12)
    groups, chunks = cout // (16 * nt), cin // 32
    out = np.zeros((groups, chunks, 13, 2 * nt, 64, 4), np.float32)
    lane = np.arange(64)
    g, j = lane >> 4, lane & 15
    for t in range(8):
        ci_in_chunk = np.array([_ch(32, int(gg), t) for gg in g])           # [64]
        for n in range(nt):
            idx = t * nt + n
            for grp in range(groups):
                co = grp * 16 * nt + nt * j + n                             # [64]
                for c in range(chunks):
                    out[grp, c, :12, idx // 4, :, idx % 4] = U[co, c * 32 + ci_in_chunk, :].T
    return torch.from_numpy(out)
