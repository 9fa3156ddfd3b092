"""Index logic of csrc/mixer_fused.hip's mixer_token_kernel, restated lane by lane in numpy (no GPU): the parked-operand addresses of
phase 1, the 16 x 16 x 32 fragment maps of mma.h (A: row = lane & 15, k = (lane >> 4) * 8 + [0, 8); B: column = lane & 15, same k;
D: column = lane & 15, rows (lane >> 4) * 4 + [0, 4)), the k-slot permutation of the in-lane re-pack between the two products, the
slice staging of W1 / W2 and the accumulator -> (token, channel) map of the epilogue -- against the closed form
    y = x + (gelu(LN(x)^T W1^T + b1) W2^T + b2)^T            (mlp_mixer.py:47).
Also the host-side operand preparations of mi355attn.functional that feed the kernel."""
from math import erf

import numpy as np
import torch

N, NK, KS, NT, NP = 196, 224, 7, 13, 208
PU, P2 = NK + 8, 36
W1S, W2S = 32 * PU, NP * P2
STAGE, REGION = W1S + W2S, 32 * PU
_gelu = np.vectorize(lambda v: 0.5 * v * (1 + erf(v / 2 ** 0.5)))


def _mma(a, b, c):
    A, B = np.zeros((16, 32)), np.zeros((32, 16))
    for lane in range(64):
        l15, g = lane & 15, lane >> 4
        A[l15, 8 * g:8 * g + 8] = a[lane]
        B[8 * g:8 * g + 8, l15] = b[lane]
    D = A @ B
    out = c.copy()
    for lane in range(64):
        l15, g = lane & 15, lane >> 4
        out[lane] += D[4 * g:4 * g + 4, l15]
    return out


def _emulate_wave(x, stats, lnw, lnb, w1f, w2f, b1, b2, T, half_id, wave):
    C = x.shape[1]
    cw = half_id * 256 + wave * 32
    region = np.zeros(REGION)
    for lane in range(64):                                           # phase 1: token pairs as 32-bit words, channel-major rows
        cq, tk = lane & 7, lane >> 3
        for it in range(14):
            p = it * 8 + tk
            ha, hc = np.zeros(4), np.zeros(4)
            if 2 * p < N:
                ch = slice(cw + cq * 4, cw + cq * 4 + 4)
                st = stats[4 * p:4 * p + 4]
                ha = (x[2 * p, ch] - st[0]) * st[1] * lnw[ch] + lnb[ch]
                hc = (x[2 * p + 1, ch] - st[2]) * st[3] * lnw[ch] + lnb[ch]
            for e in range(4):
                word = (cq * 4 + e) * (PU // 2) + p
                region[2 * word], region[2 * word + 1] = ha[e], hc[e]
    xb = np.zeros((2, KS, 64, 8))
    for lane in range(64):
        l15, g = lane & 15, lane >> 4
        for tt in range(2):
            for ks in range(KS):
                o0 = (tt * 16 + l15) * PU + ks * 32 + g * 8
                xb[tt, ks, lane] = region[o0:o0 + 8]
    W1C, W2C = 32 * (NK // 8), NP * 4
    o = np.zeros((2, NT, 64, 4))
    for kb in range(T // 32):
        stage = np.zeros(STAGE)
        for t in range(512):                                         # fetch / commit of one slice by the 512 threads
            for i in range(2):
                q = t + 512 * i
                if q < W1C:
                    src, dst = (kb * 32 + q // (NK // 8)) * NK + (q % (NK // 8)) * 8, (q // (NK // 8)) * PU + (q % (NK // 8)) * 8
                    stage[dst:dst + 8] = w1f[src:src + 8]
                if q < W2C:
                    src, dst = (kb * NP + (q >> 2)) * 32 + (q & 3) * 8, W1S + (q >> 2) * P2 + (q & 3) * 8
                    stage[dst:dst + 8] = w2f[src:src + 8]
        s = np.zeros((2, 2, 64, 4))
        for h2 in range(2):
            for lane in range(64):
                g = lane >> 4
                s[:, h2, lane] = b1[kb * 32 + h2 * 16 + g * 4:kb * 32 + h2 * 16 + g * 4 + 4]
            for ks in range(KS):
                wf = np.zeros((64, 8))
                for lane in range(64):
                    l15, g = lane & 15, lane >> 4
                    o0 = (h2 * 16 + l15) * PU + ks * 32 + g * 8
                    wf[lane] = stage[o0:o0 + 8]
                for tt in range(2):
                    s[tt, h2] = _mma(wf, xb[tt, ks], s[tt, h2])
        pf = np.zeros((2, 64, 8))
        for tt in range(2):
            pf[tt, :, :4], pf[tt, :, 4:] = _gelu(s[tt, 0]), _gelu(s[tt, 1])
        for nt in range(NT):
            vf = np.zeros((64, 8))
            for lane in range(64):
                l15, g = lane & 15, lane >> 4
                wr = W1S + (nt * 16 + l15) * P2 + g * 4
                vf[lane, :4], vf[lane, 4:] = stage[wr:wr + 4], stage[wr + 16:wr + 20]
            for tt in range(2):
                o[tt, nt] = _mma(pf[tt], vf, o[tt, nt])
    y = np.full((N, 32), np.nan)
    for lane in range(64):                                           # epilogue: 4 consecutive channels of one token per accumulator
        l15, g = lane & 15, lane >> 4
        for nt in range(NT):
            n = nt * 16 + l15
            if n >= N:
                continue
            for tt in range(2):
                ch = slice(cw + tt * 16 + g * 4, cw + tt * 16 + g * 4 + 4)
                y[n, tt * 16 + g * 4:tt * 16 + g * 4 + 4] = o[tt, nt, lane] + b2[n] + x[n, ch]
    return cw, y


def test_mixer_token_kernel_index_logic_lane_by_lane():
    rng = np.random.default_rng(0)
    C, T = 512, 64                                                   # two hidden blocks: both stages of the double buffer
    x = rng.standard_normal((N, C))
    lnw, lnb = rng.standard_normal(C), rng.standard_normal(C)
    W1, b1 = rng.standard_normal((T, N)) / 14, rng.standard_normal(T)
    W2, b2 = rng.standard_normal((N, T)) / 8, rng.standard_normal(N)
    mean, rstd = x.mean(1), 1 / np.sqrt(x.var(1) + 1e-5)
    stats = np.stack([mean, rstd], 1).reshape(-1)
    w1p = np.zeros((T, NK)); w1p[:, :N] = W1
    w2s = np.zeros((T // 32, NP, 32))
    for kb in range(T // 32):
        w2s[kb, :N, :] = W2[:, kb * 32:(kb + 1) * 32]
    u = (x - mean[:, None]) * rstd[:, None] * lnw + lnb
    ref = x + (_gelu(u.T @ W1.T + b1) @ W2.T + b2).T
    for half_id, wave in ((0, 0), (1, 5)):
        cw, y = _emulate_wave(x, stats, lnw, lnb, w1p.reshape(-1), w2s.reshape(-1), b1, b2, T, half_id, wave)
        assert not np.isnan(y).any(), "an output element was never written"
        assert np.abs(y - ref[:, cw:cw + 32]).max() < 1e-12


def test_host_side_operands_of_the_fused_token_mixing():
    import mi355attn  # noqa: F401  (package import only: no library call below)
    from mi355attn import functional as F
    w2 = torch.randn(196, 256)
    s = F.weight16_slices(w2, 208, 1)
    assert s.shape == (8, 208, 32) and s.dtype == torch.float16 and s.is_contiguous()
    for kb in (0, 3, 7):
        assert torch.equal(s[kb, :196, :], w2[:, kb * 32:(kb + 1) * 32].half())
        assert bool((s[kb, 196:, :] == 0).all())
    assert F.weight16_slices(w2, 208, 1) is s, "cached with the parameter"
    w1 = torch.randn(256, 196)
    p = F.weight16_padk(w1, 224, 2)
    assert p.shape == (256, 224) and p.dtype == torch.bfloat16
    assert torch.equal(p[:, :196], w1.bfloat16()) and bool((p[:, 196:] == 0).all())
