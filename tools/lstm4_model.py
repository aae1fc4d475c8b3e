"""Lane-level CPU model of the four-sequence LSTM sweeps (csrc/lstm.hip: lstm_fwd4_kernel / lstm_bwd4_kernel), under the operand
layout ASSUMED there for v_mfma_f32_4x4x1_16b_f32 (lane l: A[block l/4][row l%4], B[block l/4][column l%4], D[block l/4][row v]
[column l%4] in register v).  It executes the kernels' index arithmetic lane by lane -- both roles of a lane, the exchange scratch,
the panels' strides, the ragged last workgroup -- and compares with the step-by-step restatement the kernel tests use
(tests/emulator.py).  What it cannot check is the assumption itself: tools/mfma4x4_probe.hip does that on the device.

    python tools/lstm4_model.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "dnn-based_source_separation_amd", "src")]
from emulator import EmuBackend          # noqa: E402

NS4, XCH = 4, 80


def mfma4(a, b, c):
    """one v_mfma_f32_4x4x1_16b_f32 on a wave: a, b (64,), c (64, 4) -> d (64, 4)"""
    d = c.clone()
    for lane in range(64):
        blk, col = lane // 4, lane % 4
        for v in range(4):
            d[lane, v] += a[4 * blk + v] * b[4 * blk + col]          # A[blk][row v] comes from lane 4 blk + v, B[blk][col] from this lane
    return d


def fwd4(xg, whh, nseq, L, H, reverse):
    hout = torch.full((nseq, L, H), float("nan"), dtype=xg.dtype)
    gates = torch.full((nseq, L, 4 * H), float("nan"), dtype=xg.dtype)
    cst = torch.full((nseq, L, H), float("nan"), dtype=xg.dtype)
    HS, nw = H + 4, H // 16
    X = xg.reshape(nseq, L, 4 * H)
    for wg in range((nseq + NS4 - 1) // NS4):
        seq0 = wg * NS4
        hs = [torch.zeros(NS4 * HS, dtype=xg.dtype), torch.zeros(NS4 * HS, dtype=xg.dtype)]
        c = torch.zeros(nw, 64, dtype=xg.dtype)
        for step in range(L):
            t = L - 1 - step if reverse else step
            cur = step & 1
            for w in range(nw):
                lanes = torch.arange(64)
                ul, hi, sj = lanes % 16, lanes // 16, lanes % 4
                u = 16 * w + ul
                acc = torch.zeros(64, 4, dtype=xg.dtype)
                for k in range(H):
                    a = hs[cur][sj * HS + k]
                    b = whh[hi * H + u, k]
                    acc = mfma4(a, b, acc)
                ex = torch.full((NS4 * XCH,), float("nan"), dtype=xg.dtype)
                for v in range(NS4):
                    s = min(seq0 + v, nseq - 1)
                    pre = acc[:, v] + X[s, t, hi * H + u]
                    act = torch.where(hi == 2, torch.tanh(pre), torch.sigmoid(pre))
                    ex[v * XCH + hi * 16 + ul] = act
                    if seq0 + v < nseq:
                        gates[s, t, hi * H + u] = act
                gi, gf, gg, go = (ex[hi * XCH + g * 16 + ul] for g in range(4))
                c[w] = gf * c[w] + gi * gg
                hn = go * torch.tanh(c[w])
                hs[cur ^ 1][hi * HS + u] = hn
                for lane in range(64):
                    s = seq0 + int(hi[lane])
                    if s < nseq:
                        hout[s, t, int(u[lane])] = hn[lane]
                        cst[s, t, int(u[lane])] = c[w][lane]
    return hout, gates, cst


def bwd4(dho, gates, cst, whh, nseq, L, H, reverse):
    dxg = torch.full((nseq, L, 4 * H), float("nan"), dtype=dho.dtype)
    HG = H + 16
    DS, nw = 4 * HG + 4, H // 16
    for wg in range((nseq + NS4 - 1) // NS4):
        seq0 = wg * NS4
        dhr = torch.zeros(nw, 64, dtype=dho.dtype)
        dc = torch.zeros(nw, 64, dtype=dho.dtype)
        for step in range(L):
            t = step if reverse else L - 1 - step
            tp = t + 1 if reverse else t - 1
            das = torch.full((NS4 * DS,), float("nan"), dtype=dho.dtype)
            lanes = torch.arange(64)
            ul, hi, sj = lanes % 16, lanes // 16, lanes % 4
            for w in range(nw):
                u = 16 * w + ul
                s = torch.clamp(seq0 + hi, max=nseq - 1)
                i, f, g, o = (gates[s, t, k * H + u] for k in range(4))
                cc = cst[s, t, u]
                cp = cst[s, tp, u] if 0 <= tp < L else torch.zeros(64, dtype=dho.dtype)
                dh = dho[s, t, u] + dhr[w]
                tc = torch.tanh(cc)
                dcc = dh * o * (1 - tc * tc) + dc[w]
                da = [dcc * g * i * (1 - i), dcc * cp * f * (1 - f), dcc * i * (1 - g * g), dh * tc * o * (1 - o)]
                dc[w] = dcc * f
                for k in range(4):
                    das[hi * DS + k * HG + u] = da[k]
                    for lane in range(64):
                        if seq0 + int(hi[lane]) < nseq:
                            dxg[int(s[lane]), t, k * H + int(u[lane])] = da[k][lane]
            for w in range(nw):
                u = 16 * w + ul
                acc = torch.zeros(64, 4, dtype=dho.dtype)
                for m in range(H):
                    a = das[sj * DS + hi * HG + m]
                    b = whh[hi * H + m, u]
                    acc = mfma4(a, b, acc)
                ex = torch.full((NS4 * XCH,), float("nan"), dtype=dho.dtype)
                for v in range(NS4):
                    ex[v * XCH + hi * 16 + ul] = acc[:, v]
                dhr[w] = sum(ex[hi * XCH + q * 16 + ul] for q in range(4))
    return dxg


def main():
    torch.manual_seed(0)
    E = EmuBackend()
    worst = 0.0
    for H, nseq, L in ((16, 5, 4), (32, 9, 3), (16, 4, 2)):
        for reverse in (0, 1):
            xg = torch.randn(nseq, L, 4 * H, dtype=torch.float64)
            whh = 0.3 * torch.randn(4 * H, H, dtype=torch.float64)
            h, gt, cs = (torch.empty(nseq, L, n, dtype=torch.float64) for n in (H, 4 * H, H))
            E.lstm_fwd(xg, whh, h, gt, cs, nseq, L, H, reverse)
            h4, g4, c4 = fwd4(xg, whh, nseq, L, H, reverse)
            dho = torch.randn(nseq, L, H, dtype=torch.float64)
            dx = torch.empty(nseq, L, 4 * H, dtype=torch.float64)
            E.lstm_bwd(dho, gt, cs, whh, dx, nseq, L, H, reverse)
            dx4 = bwd4(dho, gt, cs, whh, nseq, L, H, reverse)
            errs = [(a - b).abs().max().item() for a, b in ((h, h4), (gt, g4), (cs, c4), (dx, dx4))]
            assert not any(e != e for e in errs), "an output element was never written"
            worst = max(worst, *errs)
            print("H={} nseq={} L={} reverse={}: max |diff| h {:.1e} gates {:.1e} c {:.1e} dxg {:.1e}".format(H, nseq, L, reverse, *errs))
    assert worst < 1e-12, worst
    print("index arithmetic of the four-sequence sweeps agrees with the restatement (given the assumed MFMA layout)")


if __name__ == "__main__":
    main()
