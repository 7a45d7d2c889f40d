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
BASES = ["WS", "PARAMS", "GRADS", "BUFS", "X", "OUT", "DOUT", "NOISE", "WGS", "CONST", "Y", "AUX", "WPACK", "DX"]
BASE = {n: i for i, n in enumerate(BASES)}

# ---- statistics replicas -------------------------------------------------------------------------
# Per-channel sums are accumulated with f64 atomics, which execute at the memory side: thousands of
# adds onto one address serialise (a 24-channel layer at 128x128 was 40x off its roofline).  Every
# STATS / STATS2 tensor is therefore [NREP][2][C]; a workgroup adds into replica (its id % NREP)
# and the finalize stages sum the replicas.
def stats_replicas(C: int) -> int:
    return max(1, min(64, 4096 // max(C, 1)))


# ---- prologue / activation codes -----------------------------------------------------------
PRO_NONE, PRO_AFFINE, PRO_SILU, PRO_RELU, PRO_GELU = 0, 1, 2, 3, 4   # v' = act(scale[c]*v + shift[c]); GELU = exact erf form
ACT_NONE, ACT_SILU, ACT_RELU, ACT_GELU = 0, 2, 3, 4                   # same numbering as PRO_*
ACT_MUL = 5                                                            # ACT_BWD only: multiply by X instead of act'(X)
MODE_CONV, MODE_CONVT_SCATTER, MODE_GATHER2X2 = 0, 1, 2
FLAG_SIDE, FLAG_JOIN = 1, 2      # S2kOp.flags (see s2k_program_run: side-stream fork / join)
# S2kOp.flags of CONV / WGRAD, bf16-MIXED plans only: the stage MAY round its two MFMA operands to bf16 (f32 accumulate, f32
# epilogue; csrc/conv_bf16.hip, wgrad_bf16.hip) when the shape is one of the bf16 kernels'; CONV then reads the bf16 weight
# copy WTB that WEIGHT_PACK wrote (BF16_BASE).  Without the flag every stage computes in exact f32 (the parity path).
FLAG_BF16 = 4
FLAG_Q4 = 16        # CONV (f32 plans): WTB holds the f32 "quad" copy of WT that WEIGHT_PACK wrote (Q4_BASE): [KP / 8][MP][8] with the eight
                    # channels of a group in the order (k & 1) * 4 + (k >> 1) - the A-operand layout of csrc/conv_q4.hip
FLAG_RES_GELU_GRAD = 32   # CONV with RES: Y = (conv + bias) * gelu'(RES) instead of + RES - the backward of "GELU then Linear" in one stage
                    # (fc2's data gradient times gelu'(fc1 output): timm Mlp, prithvi.py:162-183), f32 plans; the producer / consumer and
                    # generic kernels implement it, the other CONV kernels decline such a stage
FLAG_DMA = 8        # CONV: take the LDS-DMA ring kernel (csrc/conv_dma.hip) for every shape it supports, not only where its launcher's
                    # measured routing rule sends a stage (tests cover all of its tiles this way; plans leave the choice to the launcher)

# BN_FINALIZE folded into the first consumer of its {scale, shift} (a 5 us launch per BatchNorm otherwise: 126 per U-Net step):
# the consumer's BNV becomes an OUTPUT computed from FSTATS (NREP replicas of {sum, sumsq}[C], FCOUNT elements per channel) with
# FGAMMA / FBETA, and FRM / FRV receive the momentum update - the arithmetic of BN_FINALIZE with TRAIN = 1.
FOLD_T, FOLD_N, FOLD_D, FOLD_F = ["FSTATS", "FGAMMA", "FBETA", "FRM", "FRV"], ["FCOUNT"], ["FNREP"], ["FEPS", "FMOM"]
FOLD_KINDS = ("DWCONV_FWD", "SE_POOL", "BN_RESIDUAL")

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
    # BF16_BASE > 0 (bf16-mixed plans): additionally DST_bytes[BF16_BASE + 2*dst_off ...] receives every entry as bf16 in the
    # fragment order of the bf16 MFMA kernels, [KP/8][T][MP][8]: element ((kc/8 * T + tap) * MP + m) * 8 + kc % 8
    # Q4_BASE > 0 (f32 plans): additionally DST_bytes[Q4_BASE + 4*dst_off ...] receives every 1x1 entry (T = 1) in the quad layout
    # [KP/8][MP][8]: element ((kc >> 3) * MP + m) * 8 + (kc & 1) * 4 + ((kc >> 1) & 3)   (csrc/conv_q4.hip; FLAG_Q4 stages read it as WTB)
    "WEIGHT_PACK": (["TABLE", "SRC", "DST"], ["TOTAL", "BF16_BASE", "Q4_BASE"], ["N_ENTRIES"], []),
    # Implicit-GEMM convolution on f32 MFMA (fwd conv / convT fwd / conv dgrad / convT dgrad):
    #   Y[b][m][yo][xo] (+)= BIAS[m] + sum_{c,ky,kx} Wv[m][c][ky][kx] * Xpro[b][c][yo*S+ky-PT][xo*S+kx-PL]
    # with Wv[m][c][tap] = WT[m*W_SM + c*W_SK + (FLIP ? T-1-tap : tap)*W_ST] (the HIP kernel requires the
    # WEIGHT_PACK layout: W_SM = 1, W_ST = MP, W_SK = T*MP, FLIP = 0); c runs over X1's C1
    # channels then X2's C2 (channel concat without materialising it); Xpro = prologue(X, BNV, GATE).
    # MODE_CONVT_SCATTER: rows m = (co,dy,dx), stored to Y[b][co][2y+dy][2x+dx] (ConvTranspose k2 s2).
    # MODE_GATHER2X2: pseudo-channel k = (co,dy,dx) of X1 reads X1[b][co][2y+dy][2x+dx].
    # STATS (double [2][M]) accumulates sum(Y), sum(Y^2) per row for train-mode BatchNorm.
    # RES (same layout as Y): added to the result (the transformer's residual stream: x' = x + proj(...)).
    # A Linear over feature-major tokens [B][C][L] is this stage with H = 1, W = L.
    # SCRATCH (optional, >= 8 * B*YC*HO*WO floats): lets the kernel cut a long reduction over few output tiles into
    # split-K partials that a tail kernel adds in a fixed order (deep 8x8 / 16x16 layers: 160 tiles cannot fill 256 CUs).
    # WTB (with FLAG_BF16): the bf16 copy of WT written by WEIGHT_PACK (BF16_BASE)
    # X1_BF16 (bf16-mixed plans, 1x1 stages with FLAG_BF16, PRO1 = NONE, C2 = 0): X1 is stored as bf16 [B][C1][HW] - the output of a
    # BN_BWD_APPLY with OUT_BF16.  The stage would round exactly these values to bf16 when it builds its MFMA operand, so results are
    # bit-identical to reading the f32 tensor; the apply pass writes, and both consumers read, half the bytes.
    "CONV": (["X1", "BNV1", "GATE1", "X2", "BNV2", "WT", "BIAS", "Y", "STATS", "RES", "SCRATCH", "WTB"], [],
             ["B", "C1", "C2", "H", "W", "M", "KH", "KW", "STRIDE", "PAD_T", "PAD_L", "HO", "WO",
              "PRO1", "PRO2", "MODE", "W_SM", "W_SK", "W_ST", "FLIP", "BETA", "YC", "NREP", "X1_BF16"], []),
    # Weight gradient on f32 MFMA, K = pixels:
    #   WGS[tap][m][c] += sum_{b,yo,xo} Ppro[b][m][yo][xo] * Qpro[b][c][yo*S+ky-PT][xo*S+kx-PL]
    # (MODE_GATHER2X2: Q tap (dy,dx) reads Q[b][c][2y+dy][2x+dx]).  Scratch layout [T][M][CTOT]
    # keeps the float atomics 128-B contiguous; WGRAD_FINALIZE folds it into [M][C][T] grads.
    "WGRAD": (["P", "BNVP", "GATEP", "Q", "BNVQ", "GATEQ", "WGS"], [],
              ["B", "M", "C", "CTOT", "H", "W", "KH", "KW", "STRIDE", "PAD_T", "PAD_L", "HO", "WO",
               "PROP", "PROQ", "MODE", "P_BF16"], []),     # P_BF16: as CONV.X1_BF16, for P (1x1, PROP = NONE)
    # GRADS[off + (m*C + c)*T + t] += WGS[off + (t*M + m)*C + c] for every TABLE entry {off, M, C, T, start}
    "WGRAD_FINALIZE": (["TABLE", "WGS", "GRADS"], ["TOTAL"], ["N_ENTRIES"], []),
    # depthwise KxK, TF-SAME pads, prologue on X, BN stats of Y
    # FSTATS given (training): BN_FINALIZE of the INPUT's BatchNorm is folded into this stage (see FOLD_T below) - every
    # workgroup derives {scale, shift} of its channels from FSTATS itself, the first one of a channel also writes BNV and
    # updates the running statistics
    "DWCONV_FWD": (["X", "BNV", "WT", "Y", "STATS"] + FOLD_T, FOLD_N,
                   ["B", "C", "H", "W", "K", "STRIDE", "PAD_T", "PAD_L", "HO", "WO", "PRO", "NREP"] + FOLD_D, FOLD_F),
    # G[b][c][iy][ix] (+)= (sum_taps W*DY) * act'(u), u = scale*XRAW+shift; STATS2 += {sum G, sum G*xhat}
    "DWCONV_DGRAD": (["DY", "WT", "XRAW", "BNV", "G", "STATS2"], [],
                     ["B", "C", "H", "W", "K", "STRIDE", "PAD_T", "PAD_L", "HO", "WO", "PRO", "BETA", "NREP"], []),
    # DW[c][ky][kx] += sum DY[b][c][yo][xo] * Xpro[b][c][yo*S+ky-PT][xo*S+kx-PL]
    "DWCONV_WGRAD": (["DY", "X", "BNV", "DW"], [],
                     ["B", "C", "H", "W", "K", "STRIDE", "PAD_T", "PAD_L", "HO", "WO", "PRO"], []),
    # BNV = {scale, shift, mean, invstd}[C]; TRAIN: from STATS + running-stat update; else from RM/RV
    "BN_FINALIZE": (["STATS", "GAMMA", "BETA", "RM", "RV", "BNV"], ["COUNT"], ["C", "TRAIN", "NREP"], ["EPS", "MOM"]),
    # POOL[b][c] = mean_hw act(scale*Y+shift)
    "SE_POOL": (["Y", "BNV", "POOL"] + FOLD_T, FOLD_N, ["B", "C", "HW", "PRO"] + FOLD_D, FOLD_F),
    # HPRE = W1 pool + B1; GATE = sigmoid(W2 silu(HPRE) + B2)
    "SE_FC": (["POOL", "W1", "B1", "W2", "B2", "HPRE", "GATE"], [], ["B", "C", "CSQ"], []),
    # DGATE is overwritten with d(pre-sigmoid), HPRE with d(pre-SiLU); HS = scratch [B][CSQ] for silu(HPRE).
    # DW1 / DB1 / DW2 / DB2 null: the parameter gradients are left to SE_FC_WGRAD (which may run on the side stream)
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
    # COEF null: the FINALIZE arithmetic is done inside (from STATS2 / GAMMA / COUNT / NREP; DGAMMA, DBETA += the sums)
    # ACT / MULBC / ADDBC given (COEF null): GP is the raw upstream gradient and g' = (GP*MULBC + ADDBC*ADDSCALE) * act'(u) is
    # recomputed per element (the sums in STATS2 then come from SE_BN_SUMS / SE_BN_COMBINE instead of BN_BWD_REDUCE)
    # EVAL: BatchNorm ran on its running statistics (module.eval()): mean / invstd are constants, so dY = A*g' only (the
    # batch-statistics terms Bq, Cq vanish); DGAMMA / DBETA are the same sums
    # PS given (recomputing form): the sums are combined here from SE_BN_SUMS's plane sums (SE_BN_COMBINE's arithmetic, done by
    # every wave for its channel: 6*B values from L2) and STATS2 is not read - one launch less per depthwise BatchNorm
    "BN_BWD_APPLY": (["GP", "Y", "BNV", "COEF", "DY", "STATS2", "GAMMA", "DGAMMA", "DBETA", "MULBC", "ADDBC", "PS"], ["COUNT"],
                     ["B", "C", "HW", "NREP", "ACT", "EVAL", "OUT_BF16"], ["ADDSCALE"]),     # OUT_BF16: DY is written as bf16 (RNE), HW % 4 == 0
    # XOUT = (scale*Y+shift) * dcs[b] + IDENT
    "BN_RESIDUAL": (["Y", "BNV", "IDENT", "NOISE", "XOUT"] + FOLD_T, FOLD_N, ["B", "C", "HW"] + FOLD_D, ["KEEP"] + FOLD_F),
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
    # ---- Prithvi MAE-ViT / segmentation head (activations are feature-major [B][C][L]: tokens play the role of pixels) ----
    # LayerNorm over the channel axis of [B][C][HW] (token LayerNorm with HW = L; Norm2d of the neck):
    #   MR[b][hw] = {mean, rstd};  Y = (X - mean) * rstd * GAMMA[c] + BETA[c]
    "CHAN_LN_FWD": (["X", "GAMMA", "BETA", "Y", "MR"], [], ["B", "C", "HW"], ["EPS"]),
    # DX (+)= rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)), g = DY * GAMMA;  DGAMMA[c] += sum DY * xhat;  DBETA[c] += sum DY
    # DXIN (with ACCUM): the accumulation is OUT OF PLACE, DX = DXIN + ... - the residual-stream gradient of a transformer block gets
    # a fresh buffer at every update, so the weight-gradient stages still reading the old one on the side stream are never
    # overwritten (no stream join needed).  DSUM[c] += sum_{b,hw} DX (the new values): the bias gradient of the Linear whose
    # output gradient DX is (attn.proj / mlp.fc2 of the transformer blocks) - saves a CHANNEL_SUM pass over DX per Linear.
    "CHAN_LN_BWD": (["DY", "X", "MR", "GAMMA", "DX", "DGAMMA", "DBETA", "DXIN", "DSUM"], [], ["B", "C", "HW", "ACCUM"], []),
    # G[i] *= act'(X[i])
    # ACT = ACT_MUL: G[i] *= X[i]  (the dropout gate of EfficientNet's classifier head in the backward)
    "ACT_BWD": (["G", "X"], ["COUNT"], ["ACT"], []),
    # Y[i] = act(X[i])   (GELU of the MLP hidden layer, materialised once: the erf polynomial costs ~25 vector instructions
    # per element, and a conv / wgrad prologue would re-evaluate it for every output-channel tile that reads the element)
    # BNV given (with C, HW): Y = act(scale[c] * X + shift[c]) on [.][C][HW] — EfficientNet.encode materialises SiLU(BN(conv_head)),
    # which the fused network only ever applies as a load prologue
    # GATE given (with BNV): Y = act(scale[c] * X + shift[c]) * GATE[b][c] - the SE-gated MBConv activation materialised once for the f32 plans,
    # whose project conv and weight gradient then read a plain tensor on the producer / consumer kernels (no 8-instruction prologue)
    "ACT_FWD": (["X", "Y", "BNV", "GATE"], ["COUNT"], ["ACT", "C", "HW"], []),
    # multi-head attention on QKV [B][3*HEADS*HD][LS] (rows q | k | v, each (head, d); LS >= L is the row stride, 0 = L):
    #   O[b][h*HD + d][i] = sum_j softmax_j(SCALE * <q_i, k_j>) * v_j[d];  LSE[b][h][i] = log sum_j exp(SCALE * <q_i, k_j>)
    # backward reads O and LSE back, uses DELTA [B][HEADS][LS] as scratch (sum_d DO * O);  O / DQKV / LSE columns L..LS-1 := 0
    "ATTN_FWD": (["QKV", "O", "LSE"], [], ["B", "HEADS", "HD", "L", "LS"], ["SCALE"]),
    "ATTN_BWD": (["QKV", "DO", "DQKV", "O", "LSE", "DELTA"], [], ["B", "HEADS", "HD", "L", "LS"], ["SCALE"]),
    # rank r of NOISE[b][l] in its row (ties: lower index first = a stable argsort): IDS_RESTORE[b][l] = r (int64),
    # MASK[b][l] = r >= KEEP (f32);  gather tables (int32): ENC_IDX[b] = {-1 (cls), index of rank 0 .. KEEP-1},
    # DEC_IDX[b] = {0, (r_l < KEEP ? 1 + r_l : -1) for l < L}
    "MAE_MASK_INDEX": (["NOISE", "IDS_RESTORE", "MASK", "ENC_IDX", "DEC_IDX"], [], ["B", "L", "KEEP"], []),
    # the decoder's gather table from a caller-supplied IDS_RESTORE (forward_decoder(x, ids_restore), prithvi.py:307-316):
    # DEC_IDX[b][0] = 0 (cls);  DEC_IDX[b][1 + l] = IDS[b][l] < KEEP ? 1 + IDS[b][l] : -1 (mask token)
    "IDS_TO_DEC_IDX": (["IDS", "DEC_IDX"], [], ["B", "L", "KEEP"], []),
    # OUT[b][c][j] = (i = IDX[b][j]) >= 0 ? IN[b][c][i] : FILL[c];  then + POS[(POS_BY_SRC ? i + POS_OFF : j) * C + c]
    # (POS is token-major, as the reference's pos_embed parameters)
    "TOKEN_GATHER": (["IN", "IDX", "FILL", "POS", "OUT"], [], ["B", "C", "LIN", "LOUT", "POS_BY_SRC", "POS_OFF", "LIN_S", "LOUT_S"], []),
    # DIN[b][c][i] = DOUT[b][c][j] for i = IDX[b][j] >= 0 (other DIN entries zero);  DFILL[c] += sum_{b, j: IDX < 0} DOUT[b][c][j]
    "TOKEN_SCATTER": (["DOUT", "IDX", "DIN", "DFILL"], [], ["B", "C", "LIN", "LOUT", "LIN_S", "LOUT_S"], []),
    # im2col of non-overlapping patches (PatchEmbed's Conv3d as a GEMM):
    #   OUT[b][((c*TUB + tt)*P + py)*P + px][(t, h, w)] = X[b][c][t*TUB + tt][h*P + py][w*P + px]
    # INVERSE: the same index map the other way round, X[...] = OUT[...] (every pixel belongs to exactly one patch): turns the
    # gradient of the patch columns into the gradient w.r.t. the images
    # INVERSE 2: X = -OUT, 3: X += OUT.  ORDER 1: row index (tt, py, px, c) — the MAE loss target's feature order (prithvi.py:236-245)
    # instead of the Conv3d weight's (c, tt, py, px).  LS / L_OFF: row stride (0 = L) and first column of OUT's patches.
    # INVERSE 4 (ORDER 1): X = the gradient w.r.t. the images through the per-patch STANDARDISED loss target (norm_pix_loss,
    # prithvi.py:341-344): with g = -OUT (d loss / d target), x = the patch of IMGS, n = PD, mu / v = its mean / unbiased variance,
    # s = sqrt(v + 1e-6):   X_j = (g_j - mean(g)) / s - (x_j - mu) * sum_i g_i (x_i - mu) / (s^3 (n - 1))
    "PATCHIFY": (["X", "OUT", "IMGS"], [], ["B", "C", "T", "H", "W", "P", "TUB", "INVERSE", "ORDER", "LS", "L_OFF"], []),
    # MAE loss (prithvi.py:333-350).  PRED is feature-major [B][PD][LP], token l in column l + L_OFF, feature order
    # (tt, py, px, c).  LOSS[0] = sum_l MASK * mean_f (PRED - target)^2 / sum MASK;  NORM_PIX: per-patch standardised target
    "MAE_LOSS_FWD": (["PRED", "IMGS", "MASK", "LOSS", "ACC"], [],
                     ["B", "C", "T", "H", "W", "P", "TUB", "LP", "L_OFF", "NORM_PIX"], []),
    "MAE_LOSS_BWD": (["PRED", "IMGS", "MASK", "ACC", "GOUT", "DPRED"], [],
                     ["B", "C", "T", "H", "W", "P", "TUB", "LP", "L_OFF", "NORM_PIX"], []),
    # Y[b][j][c] = X[b][c][j + L_OFF], j < LOUT   (feature-major -> the reference's token-major tensors at the API boundary)
    # YS > 0: rows of Y have stride YS floats and the C values go to columns Y_OFF .. Y_OFF + C - 1, every other column of the row
    # is written as zero (token-major API tensors -> feature-major rows padded to a multiple of 4 with zero padding columns)
    "TRANSPOSE_CL": (["X", "Y"], [], ["B", "C", "L", "L_OFF", "LOUT", "YS", "Y_OFF"], []),
    # HIST[t*C + p] += #{i : LABELS[i] == t, PRED[i] == p}  (int64; labels / predictions outside [0, C) are skipped):
    # the one accumulator behind confusion matrix, IoU, accuracy and F1 (train_segmentation.py:53-63,145-159) — no host sync
    "CONFUSION": (["PRED", "LABELS", "HIST"], ["COUNT"], ["C"], []),
    # GATE[i] = (U[i] >= P) / (1 - P)   (Dropout2d: one draw per (sample, channel), applied as a channel gate of the next conv)
    "DROP_GATE": (["U", "GATE"], ["COUNT"], [], ["P"]),
    # input pipeline (s2osm_dataset.py:51-71 + the albumentations Compose of s2osm_datamodule.py:75-87): for output sample b,
    # PARAMS[b] = {src tile, y0, x0, flip bits (1 = horizontal, 2 = vertical)}:
    #   X[b][c][i][j] = (float(RAW[src][c][y0 + i'][x0 + j']) - NORM[0][c]) * NORM[1][c],  i' = vflip ? S-1-i : i, j' likewise
    #   Y[b][i][j]    = LUT[LABELS[src][y0 + i'][x0 + j']]     (int64; LUT has 256 int32 entries: the CNES remap, or identity)
    # RAW int16 [NSRC][C][H][W], LABELS uint8 [NSRC][H][W]; subtract and multiply are separately rounded (numpy semantics)
    "TILE_PREP": (["RAW", "LABELS", "PARAMS", "NORM", "LUT", "X", "Y"], [], ["B", "C", "H", "W", "S", "NSRC"], []),
    # SE + BatchNorm backward of a SiLU(BN(y)) * gate activation in two passes (see csrc/ew.hip): per (b, c) plane,
    # u = scale*y + shift, a' = silu'(u), xhat = (y - mean)*invstd:
    #   DGATE = sum G*silu(u);  PS[0] = sum G*a';  PS[1] = sum a';  PS[2] = sum G*a'*xhat;  PS[3] = sum a'*xhat     (PS [4][B][C])
    "SE_BN_SUMS": (["G", "Y", "BNV", "DGATE", "PS"], [], ["B", "C", "HW", "ACT"], []),
    # STATS2[0][c] = sum_b MULBC*PS[0] + ADDBC*ADDSCALE*PS[1];  STATS2[1][c] = sum_b MULBC*PS[2] + ADDBC*ADDSCALE*PS[3]   (f64 [2][C])
    "SE_BN_COMBINE": (["PS", "MULBC", "ADDBC", "STATS2"], [], ["B", "C"], ["ADDSCALE"]),
    # Y[b][c*4 + dy*2 + dx][y][x] = X[b][c][2y + dy][2x + dx]   (H, W = the low-resolution size): the gradient of a
    # ConvTranspose2d(k2, s2) output regrouped so that its weight / data gradients are plain 1x1 contractions over 4*C channels
    "SPACE_TO_DEPTH": (["X", "Y"], [], ["B", "C", "H", "W"], []),
    # Y[b][c][S*y][S*x] = X[b][c][y][x], every other element of Y [B][C][HO][WO] zero: the gradient of a STRIDED dense conv's output
    # spread onto the input grid, so that its data gradient is a stride-1 conv with the flipped kernel (only planned when the
    # caller asks for the gradient w.r.t. the network input: the stem is the only strided dense conv)
    "UPSAMPLE_ZERO": (["X", "Y"], [], ["B", "C", "H", "W", "S", "HO", "WO"], []),
    # parameter gradients of the two SE Linears from what SE_FC_BWD left behind (DGP = its DGATE, DHP = its HPRE, HS):
    #   DW2[c][j] += sum_b DGP[b][c]*HS[b][j];  DB2 += sum_b DGP;  DW1[j][c] += sum_b DHP[b][j]*POOL[b][c];  DB1 += sum_b DHP
    "SE_FC_WGRAD": (["DGP", "HS", "DHP", "POOL", "DW1", "DB1", "DW2", "DB2"], [], ["B", "C", "CSQ"], []),
    # Y[b][(c*KH + ky)*KW + kx][yo][xo] = X[b][c][yo*S + ky - PT][xo*S + kx - PL]   (zero outside X): the patch columns of a
    # STRIDED dense conv (the stem: 13 -> 48 channels, 3x3, stride 2).  The conv and its weight gradient then are 1x1 contractions
    # over C*KH*KW pseudo-channels - the parameter's own [M][C][KH][KW] layout is that 1x1 weight - on the fast pixel-tile kernels
    # instead of the generic strided path (17 - 24 TF/s there; the weight gradient was the last, fully exposed stage of a step)
    "IM2COL": (["X", "Y"], [], ["B", "C", "H", "W", "KH", "KW", "STRIDE", "PAD_T", "PAD_L", "HO", "WO"], []),
}
# Tensor slots a stage WRITES (everything else it only reads).  Used by the planner's side-stream hazard pass (unet_plan.finish_plan:
# a main-stream stage that writes what an outstanding side-stream stage still reads must wait for the side stream first); a kind
# missing here is treated as writing every tensor it names.
WRITES: dict[str, tuple[str, ...]] = {
    "MEMSET": ("DST",), "AXPY": ("Y",), "WEIGHT_PACK": ("DST",), "CONV": ("Y", "STATS", "SCRATCH"), "WGRAD": ("WGS",),
    "WGRAD_FINALIZE": ("GRADS",), "DWCONV_FWD": ("Y", "STATS", "BNV", "FRM", "FRV"), "DWCONV_DGRAD": ("G", "STATS2"), "DWCONV_WGRAD": ("DW",),
    "BN_FINALIZE": ("RM", "RV", "BNV"), "SE_POOL": ("POOL", "BNV", "FRM", "FRV"), "SE_FC": ("HPRE", "GATE"),
    "SE_FC_BWD": ("DGATE", "HPRE", "DW1", "DB1", "DW2", "DB2", "DPOOL", "HS"), "SE_BWD_REDUCE": ("DGATE",),
    "BN_BWD_REDUCE": ("GOUT", "STATS2"), "BN_BWD_FINALIZE": ("DGAMMA", "DBETA", "COEF"), "BN_BWD_APPLY": ("DY", "DGAMMA", "DBETA"),
    "BN_RESIDUAL": ("XOUT", "BNV", "FRM", "FRV"), "CHANNEL_SUM": ("OUT",), "LOSS_FWD": ("LOSS", "ACC"), "LOSS_BWD": ("DLOGITS",),
    "ARGMAX": ("MASK",), "CHAN_LN_FWD": ("Y", "MR"), "CHAN_LN_BWD": ("DX", "DGAMMA", "DBETA", "DSUM"), "ACT_BWD": ("G",), "ACT_FWD": ("Y",),
    "ATTN_FWD": ("O", "LSE"), "ATTN_BWD": ("DQKV", "DELTA"), "MAE_MASK_INDEX": ("IDS_RESTORE", "MASK", "ENC_IDX", "DEC_IDX"),
    "IDS_TO_DEC_IDX": ("DEC_IDX",), "TOKEN_GATHER": ("OUT",), "TOKEN_SCATTER": ("DIN", "DFILL"), "PATCHIFY": ("X", "OUT"),
    "MAE_LOSS_FWD": ("LOSS", "ACC"), "MAE_LOSS_BWD": ("DPRED",), "TRANSPOSE_CL": ("Y",), "CONFUSION": ("HIST",), "DROP_GATE": ("GATE",),
    "TILE_PREP": ("X", "Y"), "SE_BN_SUMS": ("DGATE", "PS"), "SE_BN_COMBINE": ("STATS2",), "SPACE_TO_DEPTH": ("Y",), "UPSAMPLE_ZERO": ("Y",),
    "SE_FC_WGRAD": ("DW1", "DB1", "DW2", "DB2"), "IM2COL": ("Y",),
}
for _k, _w in WRITES.items():
    assert _k in OPS and all(x in OPS[_k][0] for x in _w), _k

KIND = {name: i + 1 for i, name in enumerate(OPS)}
NAME_OF = {i: name for name, i in KIND.items()}


def slot(kind: str, field: str) -> tuple[str, int]:
    t, n, d, f = OPS[kind]
    for arr, names in (("t", t), ("n", n), ("d", d), ("f", f)):
        if field in names:
            return arr, names.index(field)
    raise KeyError(f"{kind}.{field}")


for _k, (_t, _n, _d, _f) in OPS.items():
    assert len(_t) <= N_T and len(_n) <= N_N and len(_d) <= N_D and len(_f) <= N_F, _k
