// C-ABI entry points (include/s2k.h) and the native stage executor.
//
// The executor is the "runtime" of this path: a plain loop over POD stage records that enqueues
// hand-written kernels on the caller's HIP stream.  No device allocation, no synchronisation; the only
// mutable global is a mutex-guarded pool of per-device side streams leased per call — so a whole
// forward or backward is graph-capturable, costs one FFI call, and may be issued from several host
// threads / streams / devices at once.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include <stdlib.h>

#include "common.h"

namespace s2k {

static thread_local char g_err[512] = "";
thread_local int g_s2k_variant = 0;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int launch_conv(const S2kOp&, const Ctx&);
int launch_wgrad(const S2kOp&, const Ctx&);
int launch_dwconv_fwd(const S2kOp&, const Ctx&);
int launch_dwconv_dgrad(const S2kOp&, const Ctx&);
int launch_dwconv_wgrad(const S2kOp&, const Ctx&);
int launch_axpy(const S2kOp&, const Ctx&);
int launch_weight_pack(const S2kOp&, const Ctx&);
int launch_wgrad_finalize(const S2kOp&, const Ctx&);
int launch_bn_finalize(const S2kOp&, const Ctx&);
int launch_se_pool(const S2kOp&, const Ctx&);
int launch_se_fc(const S2kOp&, const Ctx&);
int launch_se_fc_bwd(const S2kOp&, const Ctx&);
int launch_se_bwd_reduce(const S2kOp&, const Ctx&);
int launch_bn_bwd_reduce(const S2kOp&, const Ctx&);
int launch_bn_bwd_finalize(const S2kOp&, const Ctx&);
int launch_bn_bwd_apply(const S2kOp&, const Ctx&);
int launch_bn_residual(const S2kOp&, const Ctx&);
int launch_channel_sum(const S2kOp&, const Ctx&);
int launch_loss_fwd(const S2kOp&, const Ctx&);
int launch_loss_bwd(const S2kOp&, const Ctx&);
int launch_argmax(const S2kOp&, const Ctx&);
int launch_chan_ln_fwd(const S2kOp&, const Ctx&);
int launch_chan_ln_bwd(const S2kOp&, const Ctx&);
int launch_act_bwd(const S2kOp&, const Ctx&);
int launch_act_fwd(const S2kOp&, const Ctx&);
int launch_attn_fwd(const S2kOp&, const Ctx&);
int launch_attn_bwd(const S2kOp&, const Ctx&);
int launch_mae_mask_index(const S2kOp&, const Ctx&);
int launch_token_gather(const S2kOp&, const Ctx&);
int launch_token_scatter(const S2kOp&, const Ctx&);
int launch_patchify(const S2kOp&, const Ctx&);
int launch_mae_loss_fwd(const S2kOp&, const Ctx&);
int launch_mae_loss_bwd(const S2kOp&, const Ctx&);
int launch_transpose_cl(const S2kOp&, const Ctx&);
int launch_drop_gate(const S2kOp&, const Ctx&);
int launch_confusion(const S2kOp&, const Ctx&);
int launch_tile_prep(const S2kOp&, const Ctx&);
int launch_se_bn_sums(const S2kOp&, const Ctx&);
int launch_se_bn_combine(const S2kOp&, const Ctx&);
int launch_space_to_depth(const S2kOp&, const Ctx&);
int launch_se_fc_wgrad(const S2kOp&, const Ctx&);
int launch_ids_to_dec_idx(const S2kOp&, const Ctx&);
int launch_upsample_zero(const S2kOp&, const Ctx&);
int launch_im2col(const S2kOp&, const Ctx&);
int launch_adam(float*, const float*, float*, float*, int64_t, double, double, double, double, double, int, hipStream_t);
int launch_mfma_selftest(const float*, const float*, float*, hipStream_t);

static int launch_memset(const S2kOp& op, const Ctx& c) {
    const int64_t ref = op.t[S2K_MEMSET_T_DST];
    const int64_t bytes = op.n[S2K_MEMSET_N_BYTES];
    if (ref < 0 || bytes <= 0) { set_error("memset: bad args"); return S2K_EINVAL; }
    const int base = (int)(ref >> 56);
    if (base >= c.n_bases || !c.bases[base]) { set_error("memset: null base %d", base); return S2K_EFAULT; }
    char* dst = static_cast<char*>(c.bases[base]) + (ref & ((1ll << 56) - 1));
    if (hipMemsetAsync(dst, 0, (size_t)bytes, c.stream) != hipSuccess) { set_error("memset: hipMemsetAsync failed"); return S2K_EHIP; }
    return S2K_OK;
}

static const char* const kNames[S2K_N_KINDS + 1] = {
    nullptr,
    "MEMSET", "AXPY", "WEIGHT_PACK", "CONV", "WGRAD", "WGRAD_FINALIZE", "DWCONV_FWD", "DWCONV_DGRAD", "DWCONV_WGRAD",
    "BN_FINALIZE", "SE_POOL", "SE_FC", "SE_FC_BWD", "SE_BWD_REDUCE", "BN_BWD_REDUCE", "BN_BWD_FINALIZE", "BN_BWD_APPLY",
    "BN_RESIDUAL", "CHANNEL_SUM", "LOSS_FWD", "LOSS_BWD", "ARGMAX", "CHAN_LN_FWD", "CHAN_LN_BWD", "ACT_BWD", "ACT_FWD",
    "ATTN_FWD", "ATTN_BWD", "MAE_MASK_INDEX", "IDS_TO_DEC_IDX", "TOKEN_GATHER", "TOKEN_SCATTER", "PATCHIFY", "MAE_LOSS_FWD",
    "MAE_LOSS_BWD", "TRANSPOSE_CL", "CONFUSION", "DROP_GATE", "TILE_PREP", "SE_BN_SUMS", "SE_BN_COMBINE", "SPACE_TO_DEPTH",
    "UPSAMPLE_ZERO", "SE_FC_WGRAD", "IM2COL"};

static int dispatch(const S2kOp& op, const Ctx& c) {
    switch (op.kind) {
        case S2K_OP_MEMSET: return launch_memset(op, c);
        case S2K_OP_AXPY: return launch_axpy(op, c);
        case S2K_OP_WEIGHT_PACK: return launch_weight_pack(op, c);
        case S2K_OP_CONV: return launch_conv(op, c);
        case S2K_OP_WGRAD: return launch_wgrad(op, c);
        case S2K_OP_WGRAD_FINALIZE: return launch_wgrad_finalize(op, c);
        case S2K_OP_DWCONV_FWD: return launch_dwconv_fwd(op, c);
        case S2K_OP_DWCONV_DGRAD: return launch_dwconv_dgrad(op, c);
        case S2K_OP_DWCONV_WGRAD: return launch_dwconv_wgrad(op, c);
        case S2K_OP_BN_FINALIZE: {
#ifdef S2K_TUNING
            static const int skip = tune_int("S2K_EXP_SKIP_BN_FINALIZE", 0);   // timing experiment only: stale scale / shift
            if (skip) return S2K_OK;
#endif
            return launch_bn_finalize(op, c);
        }
        case S2K_OP_SE_POOL: return launch_se_pool(op, c);
        case S2K_OP_SE_FC: return launch_se_fc(op, c);
        case S2K_OP_SE_FC_BWD: return launch_se_fc_bwd(op, c);
        case S2K_OP_SE_BWD_REDUCE: return launch_se_bwd_reduce(op, c);
        case S2K_OP_BN_BWD_REDUCE: return launch_bn_bwd_reduce(op, c);
        case S2K_OP_BN_BWD_FINALIZE: return launch_bn_bwd_finalize(op, c);
        case S2K_OP_BN_BWD_APPLY: return launch_bn_bwd_apply(op, c);
        case S2K_OP_BN_RESIDUAL: return launch_bn_residual(op, c);
        case S2K_OP_CHANNEL_SUM: return launch_channel_sum(op, c);
        case S2K_OP_LOSS_FWD: return launch_loss_fwd(op, c);
        case S2K_OP_LOSS_BWD: return launch_loss_bwd(op, c);
        case S2K_OP_ARGMAX: return launch_argmax(op, c);
        case S2K_OP_CHAN_LN_FWD: return launch_chan_ln_fwd(op, c);
        case S2K_OP_CHAN_LN_BWD: return launch_chan_ln_bwd(op, c);
        case S2K_OP_ACT_BWD: return launch_act_bwd(op, c);
        case S2K_OP_ACT_FWD: return launch_act_fwd(op, c);
        case S2K_OP_ATTN_FWD: return launch_attn_fwd(op, c);
        case S2K_OP_ATTN_BWD: return launch_attn_bwd(op, c);
        case S2K_OP_MAE_MASK_INDEX: return launch_mae_mask_index(op, c);
        case S2K_OP_TOKEN_GATHER: return launch_token_gather(op, c);
        case S2K_OP_TOKEN_SCATTER: return launch_token_scatter(op, c);
        case S2K_OP_PATCHIFY: return launch_patchify(op, c);
        case S2K_OP_MAE_LOSS_FWD: return launch_mae_loss_fwd(op, c);
        case S2K_OP_MAE_LOSS_BWD: return launch_mae_loss_bwd(op, c);
        case S2K_OP_TRANSPOSE_CL: return launch_transpose_cl(op, c);
        case S2K_OP_DROP_GATE: return launch_drop_gate(op, c);
        case S2K_OP_CONFUSION: return launch_confusion(op, c);
        case S2K_OP_TILE_PREP: return launch_tile_prep(op, c);
        case S2K_OP_SE_BN_SUMS: return launch_se_bn_sums(op, c);
        case S2K_OP_SE_BN_COMBINE: return launch_se_bn_combine(op, c);
        case S2K_OP_SPACE_TO_DEPTH: return launch_space_to_depth(op, c);
        case S2K_OP_SE_FC_WGRAD: return launch_se_fc_wgrad(op, c);
        case S2K_OP_IDS_TO_DEC_IDX: return launch_ids_to_dec_idx(op, c);
        case S2K_OP_UPSAMPLE_ZERO: return launch_upsample_zero(op, c);
        case S2K_OP_IM2COL: return launch_im2col(op, c);
        default: set_error("unknown stage kind %d", op.kind); return S2K_ENOSYS;
    }
}

static int check_launch(int rc, const S2kOp& op, int index) {
    if (rc != S2K_OK) {
        char tmp[400];
        snprintf(tmp, sizeof(tmp), "%s", g_err);
        set_error("op %d (%s): %s", index, (op.kind > 0 && op.kind <= S2K_N_KINDS) ? kNames[op.kind] : "?", tmp);
        return rc;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("op %d (%s): HIP launch error: %s", index, (op.kind > 0 && op.kind <= S2K_N_KINDS) ? kNames[op.kind] : "?",
                  hipGetErrorString(e));
        return S2K_EHIP;
    }
    return S2K_OK;
}

}  // namespace s2k

using namespace s2k;

extern "C" {

int s2k_abi_version(void) { return S2K_ABI_VERSION; }
size_t s2k_op_size(void) { return sizeof(S2kOp); }
const char* s2k_last_error(void) { return g_err; }
const char* s2k_kind_name(int kind) { return (kind > 0 && kind <= S2K_N_KINDS) ? kNames[kind] : nullptr; }

// Side stream for stages flagged S2K_FLAG_SIDE (weight gradients: nothing downstream in the backward chain reads them
// until the bucket's WGRAD_FINALIZE): they run concurrently with the main chain, so an MFMA-bound wgrad overlaps the
// HBM-bound BatchNorm / depthwise stages of the layers below.  Fork = event on the caller's stream the side stream waits
// for (re-armed whenever main-stream work was issued since the last fork); join = the reverse, in front of every stage
// flagged S2K_FLAG_JOIN and at the end of the call, so on return all work is ordered on the caller's stream again.
//
// Side queues are pooled PER DEVICE and leased for the duration of one s2k_program_run: two host threads, two caller
// streams or two devices in one process never share a stream or an event pair (the pool, guarded by a mutex, is the only
// mutable global of the library; a queue goes back to the pool when the call returns — by then every event it recorded has
// been waited for by the caller's stream, so the next lessee may re-record them).
struct SideQueue {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
constexpr int kMaxDevices = 64;
static std::mutex g_pool_mu;
static std::vector<SideQueue*> g_pool[kMaxDevices];

static bool side_streams_enabled() {
    static const bool on = tune_int("S2K_NO_SIDE_STREAM", 0) == 0;
    return on;
}

static SideQueue* lease_side_queue(int device) {
    if (device < 0 || device >= kMaxDevices || !side_streams_enabled()) return nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pool[device].empty()) {
            SideQueue* q = g_pool[device].back();
            g_pool[device].pop_back();
            return q;
        }
    }
    SideQueue* q = new SideQueue;   // created with `device` current (the caller holds the device guard)
    if (hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&q->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&q->join, hipEventDisableTiming) != hipSuccess) {
        if (q->fork) (void)hipEventDestroy(q->fork);
        if (q->stream) (void)hipStreamDestroy(q->stream);
        delete q;
        (void)hipGetLastError();
        return nullptr;                   // no side stream: everything runs on the caller's stream (still correct)
    }
    return q;
}

static void release_side_queue(int device, SideQueue* q) {
    if (!q) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool[device].push_back(q);
}

// Makes the device that owns `stream` current for the duration of a call (kernel launches, event / stream creation and
// hipFuncSetAttribute all act on the current device) and restores the caller's device afterwards.
struct DeviceGuard {
    int prev = -1, dev = -1;
    bool switched = false;
    explicit DeviceGuard(hipStream_t st) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        dev = prev;
        int sd = -1;
        if (st != nullptr && hipStreamGetDevice(st, &sd) == hipSuccess && sd >= 0) dev = sd;
        (void)hipGetLastError();
        if (dev != prev && dev >= 0) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched && prev >= 0) (void)hipSetDevice(prev);
    }
};

int s2k_program_run(const S2kOp* ops, int begin, int end, void* const* bases, int n_bases, void* stream) {
    if (!ops || !bases || begin < 0 || end < begin) { set_error("program_run: bad arguments"); return S2K_EINVAL; }
    hipStream_t main_st = static_cast<hipStream_t>(stream);
    DeviceGuard guard(main_st);
    Ctx c{bases, n_bases, main_st};
    bool any_side = false;
    for (int i = begin; i < end && !any_side; ++i) any_side = (ops[i].flags & S2K_FLAG_SIDE) != 0;
    SideQueue* sq = any_side ? lease_side_queue(guard.dev) : nullptr;
    bool side_busy = false;      // side work issued and not yet joined
    bool main_dirty = true;      // main-stream work issued since the last fork
    auto join = [&]() {
        if (!side_busy) return;
        (void)hipEventRecord(sq->join, sq->stream);
        (void)hipStreamWaitEvent(main_st, sq->join, 0);
        side_busy = false;
    };
    int rc = S2K_OK;
    for (int i = begin; i < end; ++i) {
        const S2kOp& op = ops[i];
        if ((op.flags & S2K_FLAG_SIDE) && sq) {
            if (main_dirty) {
                (void)hipEventRecord(sq->fork, main_st);
                (void)hipStreamWaitEvent(sq->stream, sq->fork, 0);
                main_dirty = false;
            }
            Ctx cs{bases, n_bases, sq->stream};
            rc = check_launch(dispatch(op, cs), op, i);
            side_busy = true;
        } else {
            if (op.flags & S2K_FLAG_JOIN) join();
            rc = check_launch(dispatch(op, c), op, i);
            main_dirty = true;
        }
        if (rc != S2K_OK) break;
    }
    join();
    release_side_queue(guard.dev, sq);
    return rc;
}

int s2k_op_launch(const S2kOp* op, void* const* bases, int n_bases, void* stream) {
    if (!op || !bases) { set_error("op_launch: bad arguments"); return S2K_EINVAL; }
    DeviceGuard guard(static_cast<hipStream_t>(stream));
    Ctx c{bases, n_bases, static_cast<hipStream_t>(stream)};
    return check_launch(dispatch(*op, c), *op, 0);
}

static int profile_impl(const S2kOp* ops, int begin, int end, void* const* bases, int n_bases, void* stream, float* ms_by_kind,
                        int* launches_by_kind, float* ms_per_op, int* variant_per_op);

int s2k_program_profile(const S2kOp* ops, int begin, int end, void* const* bases, int n_bases, void* stream, float* ms_by_kind,
                        int* launches_by_kind) {
    return profile_impl(ops, begin, end, bases, n_bases, stream, ms_by_kind, launches_by_kind, nullptr, nullptr);
}

int s2k_program_profile_ops(const S2kOp* ops, int begin, int end, void* const* bases, int n_bases, void* stream,
                            float* ms_per_op) {
    float ms[S2K_N_KINDS + 1];
    int cnt[S2K_N_KINDS + 1];
    return profile_impl(ops, begin, end, bases, n_bases, stream, ms, cnt, ms_per_op, nullptr);
}

int s2k_program_profile_variants(const S2kOp* ops, int begin, int end, void* const* bases, int n_bases, void* stream,
                                 float* ms_per_op, int* variant_per_op) {
    float ms[S2K_N_KINDS + 1];
    int cnt[S2K_N_KINDS + 1];
    if (!ms_per_op || !variant_per_op) { set_error("program_profile_variants: bad arguments"); return S2K_EINVAL; }
    return profile_impl(ops, begin, end, bases, n_bases, stream, ms, cnt, ms_per_op, variant_per_op);
}

static int profile_impl(const S2kOp* ops, int begin, int end, void* const* bases, int n_bases, void* stream, float* ms_by_kind,
                        int* launches_by_kind, float* ms_per_op, int* variant_per_op) {
    if (!ops || !bases || !ms_by_kind || !launches_by_kind || begin < 0 || end < begin) {
        set_error("program_profile: bad arguments");
        return S2K_EINVAL;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    DeviceGuard guard(st);
    Ctx c{bases, n_bases, st};
    const int n = end - begin;
    hipEvent_t* ev = new hipEvent_t[n + 1];
    for (int i = 0; i <= n; ++i) hipEventCreate(&ev[i]);
    int rc = S2K_OK;
    hipEventRecord(ev[0], st);
    int done = 0;
    for (int i = 0; i < n; ++i) {
        g_s2k_variant = 0;
        rc = check_launch(dispatch(ops[begin + i], c), ops[begin + i], begin + i);
        if (rc != S2K_OK) break;
        if (variant_per_op) variant_per_op[i] = g_s2k_variant;
        hipEventRecord(ev[i + 1], st);
        done = i + 1;
    }
    hipStreamSynchronize(st);
    if (rc == S2K_OK) {
        for (int k = 0; k <= S2K_N_KINDS; ++k) { ms_by_kind[k] = 0.0f; launches_by_kind[k] = 0; }
        for (int i = 0; i < done; ++i) {
            float ms = 0.0f;
            hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            const int k = ops[begin + i].kind;
            if (ms_per_op) ms_per_op[i] = ms;
            if (k > 0 && k <= S2K_N_KINDS) { ms_by_kind[k] += ms; launches_by_kind[k] += 1; }
        }
    }
    for (int i = 0; i <= n; ++i) hipEventDestroy(ev[i]);
    delete[] ev;
    return rc;
}

int s2k_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                  double weight_decay, int step, void* stream) {
    DeviceGuard guard(static_cast<hipStream_t>(stream));
    const int rc = launch_adam(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, static_cast<hipStream_t>(stream));
    if (rc != S2K_OK) return rc;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("adam: %s", hipGetErrorString(e)); return S2K_EHIP; }
    return S2K_OK;
}

int s2k_selftest_mfma(const float* a, const float* b, float* d, void* stream) {
    if (!a || !b || !d) { set_error("selftest: null pointer"); return S2K_EINVAL; }
    DeviceGuard guard(static_cast<hipStream_t>(stream));
    launch_mfma_selftest(a, b, d, static_cast<hipStream_t>(stream));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("selftest: %s", hipGetErrorString(e)); return S2K_EHIP; }
    return S2K_OK;
}

}  // extern "C"
