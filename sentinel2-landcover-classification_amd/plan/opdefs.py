"""Stage (op) descriptors of the s2k C-ABI — single source of truth.

A *program* is an array of fixed-size POD records (``S2kOp``, 256 bytes) that the native
executor (`s2k_program_run`, csrc/s2k_api.cpp) walks, launching one hand-written HIP kernel
family per record.  Python only plans (this package); no Python runs per stage at step time.

    struct S2kOp { int32 kind; int32 flags; int64 t[12]; int64 n[4]; int32 d[24]; float f[6]; }

``t`` slots are tensor references ``(base_id << 56) | byte_offset`` (or -1 = null); the bases
(workspace arena, flat params, flat grads, ...) are passed to `s2k_program_run` as raw device
pointers, so the same program is valid for any allocation.  ``tools/gen_opdefs.py`` renders the
tables below into include/s2k_ops.h; tests check the header is in sync.
"""
from __future__ import annotations

N_T, N_N, N_D, N_F = 12, 4, 24, 6
OP_BYTES = 8 + 8 * N_T + 8 * N_N + 4 * N_D + 4 * N_F
assert OP_BYTES == 256

# ---- bases -------------------------------------------------------------------------------
BASES = ["WS", "PARAMS", "GRADS", "BUFS", "X", "OUT", "DOUT", "NOISE", "WGS", "CONST", "Y", "AUX", "WPACK"]
BASE = {n: i for i, n in enumerate(BASES)}

# ---- statistics replicas -------------------------------------------------------------------------
# Per-channel sums are accumulated with f64 atomics, which execute at the memory side: thousands of
# adds onto one address serialise (a 24-channel layer at 128x128 was 40x off its roofline).  Every
# STATS / STATS2 tensor is therefore [NREP][2][C]; a workgroup adds into replica (its id % NREP)
# and the finalize stages sum the replicas.
def stats_replicas(C: int) -> int:
    return max(1, min(64, 4096 // max(C, 1)))


# ---- prologue / activation codes -----------------------------------------------------------
PRO_NONE, PRO_AFFINE, PRO_SILU, PRO_RELU = 0, 1, 2, 3          # v' = act(scale[c]*v + shift[c])
ACT_NONE, ACT_SILU, ACT_RELU = 0, 2, 3                          # same numbering as PRO_*
MODE_CONV, MODE_CONVT_SCATTER, MODE_GATHER2X2 = 0, 1, 2

# kind -> (t slots, n slots, d slots, f slots); positional
OPS: dict[str, tuple[list[str], list[str], list[str], list[str]]] = {
    # zero `BYTES` bytes at DST
    "MEMSET": (["DST"], ["BYTES"], [], []),
    # Y[i] += X[i], COUNT floats
    "AXPY": (["X", "Y"], ["COUNT"], [], []),
    # Pack weights for the implicit GEMM: for every TABLE row {src_off, dst_off, M, K, T, s_m, s_k, s_t, flip, MP, KP,
    # start}:  DST[dst_off + (kc*T + tap)*MP + m] = SRC[src_off + m*s_m + kc*s_k + (flip ? T-1-tap : tap)*s_t]
    # for m < M, kc < K, zero in the padding (MP = M rounded up to 128, KP = K rounded up to 64), so the
    # conv kernel copies A tiles with aligned 16-byte loads and needs no bounds checks.  Runs once per step.
    "WEIGHT_PACK": (["TABLE", "SRC", "DST"], ["TOTAL"], ["N_ENTRIES"], []),
    # Implicit-GEMM convolution on f32 MFMA (fwd conv / convT fwd / conv dgrad / convT dgrad):
    #   Y[b][m][yo][xo] (+)= BIAS[m] + sum_{c,ky,kx} Wv[m][c][ky][kx] * Xpro[b][c][yo*S+ky-PT][xo*S+kx-PL]
    # with Wv[m][c][tap] = WT[m*W_SM + c*W_SK + (FLIP ? T-1-tap : tap)*W_ST] (the HIP kernel requires the
    # WEIGHT_PACK layout: W_SM = 1, W_ST = MP, W_SK = T*MP, FLIP = 0); c runs over X1's C1
    # channels then X2's C2 (channel concat without materialising it); Xpro = prologue(X, BNV, GATE).
    # MODE_CONVT_SCATTER: rows m = (co,dy,dx), stored to Y[b][co][2y+dy][2x+dx] (ConvTranspose k2 s2).
    # MODE_GATHER2X2: pseudo-channel k = (co,dy,dx) of X1 reads X1[b][co][2y+dy][2x+dx].
    # STATS (double [2][M]) accumulates sum(Y), sum(Y^2) per row for train-mode BatchNorm.
    "CONV": (["X1", "BNV1", "GATE1", "X2", "BNV2", "WT", "BIAS", "Y", "STATS"], [],
             ["B", "C1", "C2", "H", "W", "M", "KH", "KW", "STRIDE", "PAD_T", "PAD_L", "HO", "WO",
              "PRO1", "PRO2", "MODE", "W_SM", "W_SK", "W_ST", "FLIP", "BETA", "YC", "NREP"], []),
    # Weight gradient on f32 MFMA, K = pixels:
    #   WGS[tap][m][c] += sum_{b,yo,xo} Ppro[b][m][yo][xo] * Qpro[b][c][yo*S+ky-PT][xo*S+kx-PL]
    # (MODE_GATHER2X2: Q tap (dy,dx) reads Q[b][c][2y+dy][2x+dx]).  Scratch layout [T][M][CTOT]
    # keeps the float atomics 128-B contiguous; WGRAD_FINALIZE folds it into [M][C][T] grads.
    "WGRAD": (["P", "BNVP", "GATEP", "Q", "BNVQ", "GATEQ", "WGS"], [],
              ["B", "M", "C", "CTOT", "H", "W", "KH", "KW", "STRIDE", "PAD_T", "PAD_L", "HO", "WO",
               "PROP", "PROQ", "MODE"], []),
    # GRADS[off + (m*C + c)*T + t] += WGS[off + (t*M + m)*C + c] for every TABLE entry {off, M, C, T, start}
    "WGRAD_FINALIZE": (["TABLE", "WGS", "GRADS"], ["TOTAL"], ["N_ENTRIES"], []),
    # depthwise KxK, TF-SAME pads, prologue on X, BN stats of Y
    "DWCONV_FWD": (["X", "BNV", "WT", "Y", "STATS"], [],
                   ["B", "C", "H", "W", "K", "STRIDE", "PAD_T", "PAD_L", "HO", "WO", "PRO", "NREP"], []),
    # G[b][c][iy][ix] (+)= (sum_taps W*DY) * act'(u), u = scale*XRAW+shift; STATS2 += {sum G, sum G*xhat}
    "DWCONV_DGRAD": (["DY", "WT", "XRAW", "BNV", "G", "STATS2"], [],
                     ["B", "C", "H", "W", "K", "STRIDE", "PAD_T", "PAD_L", "HO", "WO", "PRO", "BETA", "NREP"], []),
    # DW[c][ky][kx] += sum DY[b][c][yo][xo] * Xpro[b][c][yo*S+ky-PT][xo*S+kx-PL]
    "DWCONV_WGRAD": (["DY", "X", "BNV", "DW"], [],
                     ["B", "C", "H", "W", "K", "STRIDE", "PAD_T", "PAD_L", "HO", "WO", "PRO"], []),
    # BNV = {scale, shift, mean, invstd}[C]; TRAIN: from STATS + running-stat update; else from RM/RV
    "BN_FINALIZE": (["STATS", "GAMMA", "BETA", "RM", "RV", "BNV"], ["COUNT"], ["C", "TRAIN", "NREP"], ["EPS", "MOM"]),
    # POOL[b][c] = mean_hw act(scale*Y+shift)
    "SE_POOL": (["Y", "BNV", "POOL"], [], ["B", "C", "HW", "PRO"], []),
    # HPRE = W1 pool + B1; GATE = sigmoid(W2 silu(HPRE) + B2)
    "SE_FC": (["POOL", "W1", "B1", "W2", "B2", "HPRE", "GATE"], [], ["B", "C", "CSQ"], []),
    # DGATE is overwritten with d(pre-sigmoid), HPRE with d(pre-SiLU); HS = scratch [B][CSQ] for silu(HPRE)
    "SE_FC_BWD": (["DGATE", "GATE", "HPRE", "POOL", "W1", "W2", "DW1", "DB1", "DW2", "DB2", "DPOOL", "HS"], [],
                  ["B", "C", "CSQ"], []),
    # DGATE[b][c] = sum_hw G * act(scale*Y+shift)
    "SE_BWD_REDUCE": (["G", "Y", "BNV", "DGATE"], [], ["B", "C", "HW", "PRO"], []),
    # GOUT = (G*MULBC[b][c]*dcs[b] + ADDBC[b][c]*ADDSCALE) * act'(scale*Y+shift);
    # dcs[b] = floor(KEEP + NOISE[b]) / KEEP (drop-connect) when NOISE given;
    # STATS2 (double [2][C]) += {sum GOUT, sum GOUT*xhat}
    "BN_BWD_REDUCE": (["G", "Y", "BNV", "MULBC", "ADDBC", "NOISE", "GOUT", "STATS2"], [],
                      ["B", "C", "HW", "ACT", "NREP"], ["KEEP", "ADDSCALE"]),
    # DGAMMA += S2; DBETA += S1; COEF = {A, Bq, Cq}[C] with dY = A*g' + Bq*xhat + Cq
    "BN_BWD_FINALIZE": (["STATS2", "GAMMA", "BNV", "DGAMMA", "DBETA", "COEF"], ["COUNT"], ["C", "NREP"], []),
    "BN_BWD_APPLY": (["GP", "Y", "BNV", "COEF", "DY"], [], ["B", "C", "HW"], []),
    # XOUT = (scale*Y+shift) * dcs[b] + IDENT
    "BN_RESIDUAL": (["Y", "BNV", "IDENT", "NOISE", "XOUT"], [], ["B", "C", "HW"], ["KEEP"]),
    # OUT[c] += sum_{b,hw} G[b][c][hw]
    "CHANNEL_SUM": (["G", "OUT"], [], ["B", "C", "HW"], []),
    # per-pixel CE / focal (losses.py:24-89): LOSS[0] = value; LABELS int64; ALPHA float[C] class weights
    # MODE 0 = CE (mean over non-ignored, weighted), 1 = focal (mean/sum over all pixels)
    "LOSS_FWD": (["LOGITS", "LABELS", "ALPHA", "LOSS", "ACC"], [],
                 ["B", "C", "HW", "MODE", "IGNORE", "REDUCE_SUM"], ["GAMMA", "SMOOTH"]),
    "LOSS_BWD": (["LOGITS", "LABELS", "ALPHA", "ACC", "GOUT", "DLOGITS"], [],
                 ["B", "C", "HW", "MODE", "IGNORE", "REDUCE_SUM"], ["GAMMA", "SMOOTH"]),
    # MASK[b][hw] (int64) = argmax_c LOGITS[b][c][hw], first max wins
    "ARGMAX": (["LOGITS", "MASK"], [], ["B", "C", "HW"], []),
}
KIND = {name: i + 1 for i, name in enumerate(OPS)}


def slot(kind: str, field: str) -> tuple[str, int]:
    t, n, d, f = OPS[kind]
    for arr, names in (("t", t), ("n", n), ("d", d), ("f", f)):
        if field in names:
            return arr, names.index(field)
    raise KeyError(f"{kind}.{field}")


for _k, (_t, _n, _d, _f) in OPS.items():
    assert len(_t) <= N_T and len(_n) <= N_N and len(_d) <= N_D and len(_f) <= N_F, _k
