"""Micro-benchmark of one WGRAD / CONV stage (for rocprofv3 --pmc runs): python tools/bench_op.py wgrad3 --M 128 --C 128 --H 64"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import s2lc_amd  # noqa: E402,F401
from s2lc_amd import _lib  # noqa: E402
from s2lc_amd.plan import opdefs as D  # noqa: E402
from s2lc_amd.plan.program import Arena, Program  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["wgrad3", "wgrad1", "conv3", "conv1", "dwfwd", "dwwgrad", "dwdgrad", "copy", "lnfwd", "lnbwd"])
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--M", type=int, default=128)
    ap.add_argument("--C", type=int, default=128)
    ap.add_argument("--H", type=int, default=64)
    ap.add_argument("--pro", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rep", type=int, default=20)
    ap.add_argument("--K", type=int, default=3)
    ap.add_argument("--S", type=int, default=1)
    ap.add_argument("--nostats", action="store_true")
    ap.add_argument("--gate", action="store_true", help="conv: SE gate [B][C] on X1 (with --pro 2: the MBConv project conv's prologue)")
    ap.add_argument("--q4", action="store_true", help="conv: FLAG_Q4 + the quad weight copy (csrc/conv_q4.hip)")
    ap.add_argument("--dma", action="store_true", help="conv: FLAG_DMA (the LDS-DMA ring kernel wherever it supports the shape)")
    ap.add_argument("--bias", action="store_true")
    ap.add_argument("--beta", action="store_true", help="conv: accumulate into Y (a data gradient on top of an existing one)")
    ap.add_argument("--scratch", action="store_true")
    ap.add_argument("--zeros", action="store_true", help="all-zero operands (DVFS check: the chip holds a higher clock on trivial data)")
    ap.add_argument("--N", type=int, default=0, help="conv1: tokens per image (H=1, W=N) instead of an HxH map")
    ap.add_argument("--bf16", action="store_true", help="FLAG_BF16: the bf16 MFMA kernels (conv: WTB points at arbitrary bf16 bits - timing only)")
    a = ap.parse_args()
    B, M, C, H = a.B, a.M, a.C, a.H
    ar = Arena(D.BASE["WS"])
    prog = Program()
    k = 3 if a.what.endswith("3") else 1
    T = k * k
    if a.what == "copy":
        x = torch.randn(B, C, H, H, device="cuda"); y = torch.empty_like(x)
        for _ in range(3): y.copy_(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.iters): y.copy_(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
        print(f"copy {x.numel() * 8 / 1e6:.0f} MB: {dt * 1e3:.3f} ms  {x.numel() * 8 / dt / 1e12:.2f} TB/s")
        return
    if a.what.startswith("ln"):          # channel LayerNorm on [B][C][N] (--N tokens per sample)
        N = a.N or H * H
        X = ar.alloc("x", (B, C, N)); Y = ar.alloc("y", (B, C, N)); mr = ar.alloc("mr", (B, N, 2)); ga = ar.alloc("gamma", (C,)); be = ar.alloc("beta", (C,))
        if a.what == "lnfwd":
            prog.add("CHAN_LN_FWD", X=X, GAMMA=ga, BETA=be, Y=Y, MR=mr, B=B, C=C, HW=N, EPS=1e-6)
            flops = 8.0 * B * C * N * 1000      # "TF/s" column = TB/s of read + write
        else:
            dy = ar.alloc("dy", (B, C, N)); dxin = ar.alloc("dxin", (B, C, N)) if a.beta else None
            prog.add("CHAN_LN_BWD", DY=dy, X=X, MR=mr, GAMMA=ga, DX=Y, DGAMMA=None if a.nostats else ar.alloc("dg", (C,)),
                     DBETA=None if a.nostats else ar.alloc("db", (C,)), DXIN=dxin, DSUM=ar.alloc("ds", (C,)) if a.bias else None, B=B, C=C, HW=N, ACCUM=int(a.beta))
            flops = (16.0 if a.beta else 12.0) * B * C * N * 1000
    elif a.what.startswith("dw"):
        K, S = a.K, a.S
        HO = (H + S - 1) // S
        pad = max((HO - 1) * S + K - H, 0)
        X = ar.alloc("x", (B, C, H, H)); bnv = ar.alloc("bnv", (4, C)); Wt = ar.alloc("w", (C, K, K)); Y = ar.alloc("y", (B, C, HO, HO))
        nrep = D.stats_replicas(C)
        st = ar.alloc("st", (nrep, 2, C), "f64")
        geo = dict(B=B, C=C, H=H, W=H, K=K, STRIDE=S, PAD_T=pad // 2, PAD_L=pad // 2, HO=HO, WO=HO, PRO=a.pro)
        if a.what == "dwfwd":
            prog.add("DWCONV_FWD", X=X, BNV=bnv if a.pro else None, WT=Wt, Y=Y, STATS=None if a.nostats else st, NREP=nrep, **geo)
        elif a.what == "dwwgrad":
            prog.add("DWCONV_WGRAD", DY=Y, X=X, BNV=bnv if a.pro else None, DW=Wt, **geo)
        else:
            prog.add("DWCONV_DGRAD", DY=Y, WT=Wt, XRAW=X if a.pro else None, BNV=bnv if a.pro else None, G=X if not a.pro else ar.alloc("g", (B, C, H, H)),
                     STATS2=None if a.nostats or not a.pro else st, BETA=0, NREP=nrep, **geo)
        flops = 4.0 * B * C * (H * H + HO * HO) * 1000   # "TF/s" column = TB/s of (in + out) bytes
    elif a.what.startswith("wgrad"):
        Hh, Wd = (1, a.N) if a.N else (H, H)
        P = ar.alloc("p", (B, M, Hh, Wd)); Q = ar.alloc("q", (B, C, Hh, Wd)); bq = ar.alloc("bnv", (4, C))
        wgs = ar.alloc("wgs", (T, M, C))
        prog.add("WGRAD", P=P, BNVP=None, GATEP=None, Q=Q, BNVQ=bq if a.pro else None, GATEQ=None, WGS=wgs, B=B, M=M, C=C, CTOT=C,
                 H=Hh, W=Wd, KH=k, KW=k, STRIDE=1, PAD_T=k // 2, PAD_L=k // 2, HO=Hh, WO=Wd, PROP=0, PROQ=a.pro, MODE=0,
                 **({"_flags": D.FLAG_BF16} if a.bf16 else {}))
        flops = 2.0 * M * C * T * B * Hh * Wd
    else:
        Wd = H
        if a.N:
            H, Wd = 1, a.N
        X = ar.alloc("x", (B, C, H, Wd)); bnv = ar.alloc("bnv", (4, C)); Y = ar.alloc("y", (B, M, H, Wd))
        scr = ar.alloc("scr", (8 * B * M * H * Wd,)) if a.scratch else None
        MP, KP = (M + 127) // 128 * 128, (C + 63) // 64 * 64
        W = ar.alloc("w", (KP * T, MP)); st = ar.alloc("st", (D.stats_replicas(M), 2, M), "f64")
        flg = (D.FLAG_BF16 if a.bf16 else 0) | (D.FLAG_DMA if a.dma else 0) | ((D.FLAG_Q4 | D.FLAG_DMA) if a.q4 else 0)
        prog.add("CONV", X1=X, BNV1=bnv if a.pro else None, GATE1=ar.alloc("gate", (B, C)) if a.gate else None, X2=None, BNV2=None, WT=W, BIAS=ar.alloc("bias", (M,)) if a.bias else None, Y=Y, STATS=None if a.nostats else st,
                 SCRATCH=scr, B=B, C1=C,
                 C2=0, H=H, W=Wd, M=M, KH=k, KW=k, STRIDE=1, PAD_T=k // 2, PAD_L=k // 2, HO=H, WO=Wd, PRO1=a.pro, PRO2=0, MODE=0,
                 W_SM=1, W_SK=T * MP, W_ST=MP, FLIP=0, BETA=int(a.beta), YC=M, NREP=D.stats_replicas(M),
                 _flags=flg, **({"WTB": ar.alloc("w16", (KP * T * MP // 2,))} if a.bf16 else ({"WTB": ar.alloc("wq", (KP * T, MP))} if a.q4 else {})))
        flops = 2.0 * M * C * T * B * H * Wd
    buf = (torch.randn((ar.top + 4096) // 4, device="cuda") * (0.0 if a.zeros else 0.5)).view(torch.uint8)
    bases = _lib.Bases().set("WS", buf)
    packed = prog.pack()
    st_ = torch.cuda.current_stream().cuda_stream
    import numpy as np
    R = a.rep                       # the stage R times in ONE program: back-to-back launches from the C loop (a Python call per launch
    rep = np.concatenate([packed] * R)   # costs ~10 us of host time and hid every kernel shorter than that)
    for _ in range(3):
        _lib.run(rep, bases, st_)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        _lib.run(rep, bases, st_)
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / (a.iters * R)
    ms_ev, var = _lib.profile_variants(packed, bases, st_)     # HIP-event time of the stage alone + the kernel family it took
    fam = ("generic", "producer/consumer", "bf16", "dma-ring", "quad")[int(var[-1])] if int(var[-1]) < 5 else "-"
    print(f"{a.what} B={B} M={M} C={C} H={H} pro={a.pro}: {dt * 1e3:.3f} ms  {flops / dt / 1e12:.1f} TF/s   [{fam}; events: {float(ms_ev[-1]) * 1e3:.1f} us]")
    L = _lib.lib()
    if hasattr(L, "s2k_debug_dma_counters") and fam in ("dma-ring", "quad"):     # tuning build: in-kernel stamps of the LDS-DMA ring kernel
        import ctypes
        out = (ctypes.c_ulonglong * 8)()
        read = L.s2k_debug_q4_counters if fam == "quad" else L.s2k_debug_dma_counters
        read(out, 1)
        _lib.run(packed, bases, st_)
        torch.cuda.synchronize()
        read(out, 1)
        n = max(out[6], 1)
        print(f"   per wave ({n} waves, {out[7] / n:.1f} stages): life {out[0] / n:.0f} cyc = set-up {out[1] / n:.0f} + wait/barrier {out[2] / n:.0f} "
              f"+ issue {out[3] / n:.0f} + reads/MFMA {out[4] / n:.0f} + epilogue {out[5] / n:.0f}  (100 MHz ticks x clock ratio)")
    if hasattr(L, "s2k_debug_wg_counters"):        # tuning build: in-kernel stamps of the producer / consumer wgrad
        import ctypes
        out = (ctypes.c_ulonglong * 8)()
        L.s2k_debug_wg_counters(out, 1)
        n = max(out[2], 1)
        print(f"   per consumer wave-tile: barrier wait {out[0] / n:.0f} cyc, compute {out[1] / n:.0f} cyc; "
              f"per producer wave-tile: work {out[4] / n:.0f} cyc, barrier wait {out[3] / n:.0f} cyc  ({n} wave-tiles)")


if __name__ == "__main__":
    main()
