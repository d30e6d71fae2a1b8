// extern "C" boundary of libhvc_hip.so: argument validation + translation into the launchers.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "../../include/hvc_hip.h"
#include "hvc_kernels.h"

namespace {

thread_local char g_err[256] = "";
// Device word every dropout seed is offset by (hvc_set_seed_counter); null = seeds are taken as passed.  PROCESS-wide (one
// process per GPU): torch runs the backward of an autograd Function on its device worker thread, not on the thread that set the
// counter, and the forward and backward of one step must fold the same word into their seeds.
std::atomic<const uint32_t*> g_seed_ctr{nullptr};

__global__ void seed_counter_advance_kernel(uint32_t* ctr, uint32_t step) { *ctr += step; }

struct OptionSlot {
    const char* name;
    int def_value, min_value;
};
const OptionSlot kOptionTable[hvc::kOptCount] = {
    {"HVC_ATTN_FWD_ROWS", 0, 0},   {"HVC_ATTN_FWD_WAVES", 0, 0},  {"HVC_ATTN_BWD_WAVES", 0, 0},    {"HVC_ATTN_EXTRA_LDS", 0, 0},
    {"HVC_ATTN_PIPE", 1, 0},    {"HVC_GEMM_PERSISTENT", 1, 0}, {"HVC_GEMM_STAGGER", 0, 0},      {"HVC_GEMM_HALF_TILE", 1, 0},
    {"HVC_FP8_MX", 0, 0},          {"HVC_CONV_FORCE_ADDR64", 0, 0}, {"HVC_LOSS_FUSED", 1, 0},
};
std::atomic<int> g_options[hvc::kOptCount];
// environment -> initial values, once, while the library is being loaded (before any launch and before any other thread can call in)
const bool g_options_ready = [] {
    for (int i = 0; i < hvc::kOptCount; ++i) {
        const char* e = getenv(kOptionTable[i].name);
        int v = e ? atoi(e) : kOptionTable[i].def_value;
        if (v < kOptionTable[i].min_value) v = kOptionTable[i].min_value;
        g_options[i].store(v);
    }
    return true;
}();

int fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
int hip_result(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
bool dtype_ok(int d) { return d == HVC_F32 || d == HVC_BF16; }

uint32_t drop_threshold(float p) {
    if (p <= 0.f) return 0;
    double t = (double)p * 65536.0 + 0.5;
    if (t < 1.0) t = 1.0;
    if (t > 65535.0) t = 65535.0;
    return (uint32_t)t;
}

}  // namespace

namespace hvc {
int option(Option o) { return g_options[o].load(std::memory_order_relaxed); }
int cu_count() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        return cus;
    }();
    return n;
}
}  // namespace hvc

extern "C" {

int hvc_abi_version(void) { return HVC_ABI_VERSION; }

int hvc_set_option(const char* name, int value) {
    if (!name) return fail(HVC_E_BADARG, "set_option: null name");
    for (int i = 0; i < hvc::kOptCount; ++i)
        if (strcmp(name, kOptionTable[i].name) == 0) {
            if (value < kOptionTable[i].min_value) return fail(HVC_E_BADARG, "set_option: value below the option's minimum");
            g_options[i].store(value);
            return 0;
        }
    return fail(HVC_E_BADARG, "set_option: unknown option");
}

int hvc_get_option(const char* name, int* value) {
    if (!name || !value) return fail(HVC_E_BADARG, "get_option: null argument");
    for (int i = 0; i < hvc::kOptCount; ++i)
        if (strcmp(name, kOptionTable[i].name) == 0) {
            *value = g_options[i].load();
            return 0;
        }
    return fail(HVC_E_BADARG, "get_option: unknown option");
}
const char* hvc_last_error(void) { return g_err; }

int hvc_set_seed_counter(const uint32_t* device_counter) {
    g_seed_ctr.store(device_counter);
    return 0;
}

int hvc_clear_seed_counter_if(const uint32_t* device_counter) {
    const uint32_t* expected = device_counter;
    g_seed_ctr.compare_exchange_strong(expected, nullptr);
    return 0;
}

int hvc_seed_counter_advance(uint32_t* device_counter, uint32_t step, void* stream) {
    if (!device_counter) return fail(HVC_E_BADARG, "seed_counter_advance: null counter");
    hipLaunchKernelGGL(seed_counter_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, device_counter, step);
    return hip_result(hipGetLastError(), "seed_counter_advance");
}

int hvc_device_info(int* cu_count, int* wavefront, char* arch, int arch_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_result(e, "hipGetDevice");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return hip_result(e, "hipGetDeviceProperties");
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (wavefront) *wavefront = prop.warpSize;
    if (arch && arch_len > 0) snprintf(arch, (size_t)arch_len, "%s", prop.gcnArchName);
    return 0;
}

static int fill_attn(hvc::AttnArgs& a, const void* q, const void* k, const void* v, int B, int H, int Nq, int Nk, int D,
                     int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                     int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh,
                     float scale, float p_drop, uint64_t seed, int dtype) {
    if (!q || !k || !v) return fail(HVC_E_BADARG, "attention: null operand");
    if (B < 1 || H < 1 || Nq < 1 || Nk < 1) return fail(HVC_E_BADARG, "attention: empty dimension");
    if (D != 32 && D != 64) return fail(HVC_E_UNSUPPORTED, "attention: head dim must be 32 or 64");
    if (!dtype_ok(dtype)) return fail(HVC_E_BADARG, "attention: bad dtype");
    if (!(scale > 0.f) || !(p_drop >= 0.f) || !(p_drop < 1.f)) return fail(HVC_E_BADARG, "attention: scale must be > 0 and 0 <= p_drop < 1");
    memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.v = v;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D;
    a.q_sb = q_sb; a.q_sn = q_sn; a.q_sh = q_sh;
    a.k_sb = k_sb; a.k_sn = k_sn; a.k_sh = k_sh;
    a.v_sb = v_sb; a.v_sn = v_sn; a.v_sh = v_sh;
    a.o_sb = o_sb; a.o_sn = o_sn; a.o_sh = o_sh;
    a.do_sb = o_sb; a.do_sn = o_sn; a.do_sh = o_sh;
    a.dq_sb = q_sb; a.dq_sn = q_sn; a.dq_sh = q_sh;
    a.dk_sb = k_sb; a.dk_sn = k_sn; a.dk_sh = k_sh;
    a.dv_sb = v_sb; a.dv_sn = v_sn; a.dv_sh = v_sh;
    a.scale = scale;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.seed_ctr = g_seed_ctr.load();
    a.drop_thresh = drop_threshold(p_drop);
    a.keep_scale = 1.f / (1.f - p_drop);
    a.is_bf16 = dtype == HVC_BF16;
    const int64_t strides[] = {q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh};
    bool vec = true;
    for (int64_t s : strides) vec = vec && (s % 8 == 0);
    a.vec = vec;
    return 0;
}

int hvc_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                      int B, int H, int Nq, int Nk, int D,
                      int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                      int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh,
                      float scale, float p_drop, uint64_t seed, int dtype, void* stream) {
    hvc::AttnArgs a;
    int rc = fill_attn(a, q, k, v, B, H, Nq, Nk, D, q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh, scale, p_drop, seed, dtype);
    if (rc) return rc;
    if (!o || !lse) return fail(HVC_E_BADARG, "attention_fwd: null output");
    a.o = o; a.lse = lse;
    a.vec = a.vec && aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o);
    return hip_result(hvc::attention_launch(a, false, (hipStream_t)stream), "attention_fwd");
}

int64_t hvc_attention_fwd_fp8_workspace(int B, int H, int Nk, int D) {
    if (B < 1 || H < 1 || Nk < 1 || (D != 32 && D != 64)) return -1;
    return hvc::attention_fp8_workspace_bytes(B, H, Nk, D);
}

int hvc_attention_fwd_fp8(const void* q, const void* k, const void* v, void* o, float* lse, void* workspace,
                          int B, int H, int Nq, int Nk, int D,
                          int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                          int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh,
                          float scale, float p_drop, uint64_t seed, void* stream) {
    hvc::AttnArgs a;
    int rc = fill_attn(a, q, k, v, B, H, Nq, Nk, D, q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh, scale, p_drop, seed, HVC_BF16);
    if (rc) return rc;
    if (!o || !lse || !workspace) return fail(HVC_E_BADARG, "attention_fwd_fp8: null output / workspace");
    if (!a.vec || !aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(o) || !aligned16(workspace))
        return fail(HVC_E_UNSUPPORTED, "attention_fwd_fp8: operands must be bf16 with 16-byte addressable rows (strides multiples of 8)");
    a.o = o; a.lse = lse;
    return hip_result(hvc::attention_fp8_launch(a, workspace, (hipStream_t)stream), "attention_fwd_fp8");
}

int64_t hvc_attention_bwd_workspace(int B, int H, int Nq, int Nk, int D) {
    if (B < 1 || H < 1 || Nq < 1 || Nk < 1 || D < 1) return -1;
    const int qs = hvc::attention_bwd_qsplit(B, H, Nq, Nk);
    return (int64_t)B * H * Nq + (qs > 1 ? (int64_t)2 * qs * B * H * Nk * D : 0);
}

int hvc_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                      const float* lse, float* workspace, void* dq, void* dk, void* dv,
                      int B, int H, int Nq, int Nk, int D,
                      int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                      int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh,
                      float scale, float p_drop, uint64_t seed, int phases, int dtype, void* stream) {
    hvc::AttnArgs a;
    int rc = fill_attn(a, q, k, v, B, H, Nq, Nk, D, q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh, scale, p_drop, seed, dtype);
    if (rc) return rc;
    if (phases < 0 || phases > 7) return fail(HVC_E_BADARG, "attention_bwd: phases is a 3-bit mask");
    a.phases = phases;
    if (!o || !dout || !lse || !workspace || !dq || !dk || !dv) return fail(HVC_E_BADARG, "attention_bwd: null operand");
    a.o = const_cast<void*>(o); a.dout = dout; a.lse = const_cast<float*>(lse); a.delta = workspace;
    a.qsplit = 1;
    a.dkv_partial = workspace + (int64_t)B * H * Nq;
    a.partial_floats = hvc_attention_bwd_workspace(B, H, Nq, Nk, D) - (int64_t)B * H * Nq;
    a.dq = dq; a.dk = dk; a.dv = dv;
    a.vec = a.vec && aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o) && aligned16(dout);
    return hip_result(hvc::attention_launch(a, true, (hipStream_t)stream), "attention_bwd");
}

int hvc_gemm(const void* A, const void* B, void* C, int M, int N, int K,
             int64_t lda, int64_t ldb, int64_t ldc, int a_kmajor, int b_kmajor, float alpha,
             const float* bias, int act, void* aux, void* zsave, int64_t ldz, const float* gate,
             const float* residual, int64_t ldr, int residual_rows, int rows_per_batch, float p_drop, uint64_t seed,
             float* workspace, int64_t workspace_floats, int in_dtype, int out_dtype, void* stream) {
    if (!A || !B || !C) return fail(HVC_E_BADARG, "gemm: null operand");
    if (M < 1 || N < 1 || K < 1) return fail(HVC_E_BADARG, "gemm: empty dimension");
    if (!dtype_ok(in_dtype) || !dtype_ok(out_dtype)) return fail(HVC_E_BADARG, "gemm: bad dtype");
    if (in_dtype == HVC_F32 && out_dtype == HVC_BF16) return fail(HVC_E_UNSUPPORTED, "gemm: f32 in / bf16 out not supported");
    if (act < 0 || act > 2) return fail(HVC_E_BADARG, "gemm: bad act");
    if (act == 2 && !aux) return fail(HVC_E_BADARG, "gemm: act=2 needs aux");
    if (gate && rows_per_batch < 1) return fail(HVC_E_BADARG, "gemm: gate needs rows_per_batch");
    if (!(p_drop >= 0.f) || !(p_drop < 1.f)) return fail(HVC_E_BADARG, "gemm: bad p_drop");
    hvc::GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.a_kmajor = a_kmajor != 0; g.b_kmajor = b_kmajor != 0;
    g.alpha = alpha; g.bias = bias; g.act = act; g.aux = aux; g.zsave = zsave; g.ldz = ldz; g.gate = gate; g.residual = residual; g.ldr = ldr; g.residual_rows = residual_rows > 0 ? residual_rows : 0;
    g.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : M;
    g.seed_lo = (uint32_t)seed; g.seed_hi = (uint32_t)(seed >> 32); g.seed_ctr = g_seed_ctr.load();
    g.drop_thresh = drop_threshold(p_drop);
    g.keep_scale = 1.f / (1.f - p_drop);
    g.in_bf16 = in_dtype == HVC_BF16; g.out_bf16 = out_dtype == HVC_BF16;
    g.vec_a = aligned16(A) && (lda % 8 == 0);
    g.vec_b = aligned16(B) && (ldb % 8 == 0);
    g.workspace = workspace; g.workspace_floats = workspace ? workspace_floats : 0;
    g.vec_epi = (N % 8 == 0) && aligned16(C) && (ldc % 8 == 0) && (!aux || aligned16(aux)) &&
                (!zsave || (aligned16(zsave) && ldz % 8 == 0)) && (!residual || (aligned16(residual) && ldr % 8 == 0)) &&
                (!workspace || aligned16(workspace));
    return hip_result(hvc::gemm_launch(g, (hipStream_t)stream), "gemm");
}

int64_t hvc_gemm_workspace(int M, int N, int K) {
    if (M < 1 || N < 1 || K < 1) return -1;
    return hvc::gemm_workspace_floats(M, N, K);
}

int hvc_layernorm_fwd(const float* x, const float* gamma, const float* beta, const float* scale, const float* shift,
                      void* y, float* mean, float* rstd, int rows, int C, int rows_per_batch, float eps,
                      int out_dtype, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd) return fail(HVC_E_BADARG, "layernorm_fwd: null operand");
    if ((scale == nullptr) != (shift == nullptr)) return fail(HVC_E_BADARG, "layernorm_fwd: scale and shift go together");
    if (rows < 1 || C < 1 || C > 1024) return fail(HVC_E_UNSUPPORTED, "layernorm: 1 <= C <= 1024");
    if (rows_per_batch < 1 || rows % rows_per_batch) return fail(HVC_E_BADARG, "layernorm: rows must be a multiple of rows_per_batch");
    if (!dtype_ok(out_dtype)) return fail(HVC_E_BADARG, "layernorm: bad dtype");
    hvc::LnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.gamma = gamma; a.beta = beta; a.scale = scale; a.shift = shift; a.y = y; a.mean = mean; a.rstd = rstd;
    a.rows = rows; a.C = C; a.rows_per_batch = rows_per_batch; a.eps = eps; a.out_bf16 = out_dtype == HVC_BF16;
    return hip_result(hvc::layernorm_fwd_launch(a, (hipStream_t)stream), "layernorm_fwd");
}

int64_t hvc_layernorm_bwd_workspace(int rows, int C, int rows_per_batch) {
    if (rows < 1 || C < 1 || rows_per_batch < 1 || rows % rows_per_batch) return -1;
    const int nbatch = rows / rows_per_batch;
    return (int64_t)nbatch * hvc::layernorm_bwd_blocks_per_batch(rows_per_batch) * 4 * C;
}

int hvc_layernorm_bwd(const void* dy, const float* x, const float* gamma, const float* beta, const float* scale,
                      const float* mean, const float* rstd, const float* dres, float* dx, float* dgamma, float* dbeta,
                      float* dscale, float* dshift, float* workspace, int rows, int C, int rows_per_batch,
                      int dy_dtype, void* stream) {
    if (!dy || !x || !gamma || !beta || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace)
        return fail(HVC_E_BADARG, "layernorm_bwd: null operand");
    if (scale && (!dscale || !dshift)) return fail(HVC_E_BADARG, "layernorm_bwd: dscale/dshift required with scale");
    if (rows < 1 || C < 1 || C > 1024) return fail(HVC_E_UNSUPPORTED, "layernorm: 1 <= C <= 1024");
    if (rows_per_batch < 1 || rows % rows_per_batch) return fail(HVC_E_BADARG, "layernorm: rows must be a multiple of rows_per_batch");
    if (!dtype_ok(dy_dtype)) return fail(HVC_E_BADARG, "layernorm: bad dtype");
    hvc::LnArgs a;
    memset(&a, 0, sizeof(a));
    a.dy = dy; a.x = x; a.gamma = gamma; a.beta = beta; a.scale = scale; a.mean = const_cast<float*>(mean);
    a.rstd = const_cast<float*>(rstd); a.dres = dres; a.dx = dx; a.dgamma = dgamma; a.dbeta = dbeta;
    a.dscale = scale ? dscale : nullptr; a.dshift = scale ? dshift : nullptr; a.partial = workspace;
    a.rows = rows; a.C = C; a.rows_per_batch = rows_per_batch; a.out_bf16 = dy_dtype == HVC_BF16;
    a.blocks_per_batch = hvc::layernorm_bwd_blocks_per_batch(rows_per_batch);
    return hip_result(hvc::layernorm_bwd_launch(a, (hipStream_t)stream), "layernorm_bwd");
}

int64_t hvc_branch_bwd_workspace(int rows, int N, int rows_per_batch) {
    if (rows < 1 || N < 1 || rows_per_batch < 1 || rows % rows_per_batch) return -1;
    return (int64_t)(rows / rows_per_batch) * hvc::rowops_blocks(rows_per_batch) * 2 * N;
}

int hvc_branch_bwd(const float* dy, const void* z, const float* gate, void* dz, float* dgate, float* dbias,
                   float* workspace, int rows, int N, int rows_per_batch, float p_drop, uint64_t seed,
                   int out_dtype, void* stream) {
    if (!dy || !dz || !workspace) return fail(HVC_E_BADARG, "branch_bwd: null operand");
    if (dgate && !z) return fail(HVC_E_BADARG, "branch_bwd: dgate needs z");
    if (rows < 1 || N < 1 || rows_per_batch < 1 || rows % rows_per_batch) return fail(HVC_E_BADARG, "branch_bwd: bad shape");
    if (!dtype_ok(out_dtype)) return fail(HVC_E_BADARG, "branch_bwd: bad dtype");
    hvc::BranchArgs a;
    memset(&a, 0, sizeof(a));
    a.dy = dy; a.z = z; a.gate = gate; a.dz = dz; a.partial = workspace; a.dgate = dgate; a.dbias = dbias;
    a.rows = rows; a.N = N; a.rows_per_batch = rows_per_batch; a.blocks_per_batch = hvc::rowops_blocks(rows_per_batch);
    a.out_bf16 = out_dtype == HVC_BF16;
    if (!(p_drop >= 0.f) || !(p_drop < 1.f)) return fail(HVC_E_BADARG, "branch_bwd: bad p_drop");
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.seed_ctr = g_seed_ctr.load();
    a.drop_thresh = drop_threshold(p_drop); a.keep_scale = 1.f / (1.f - p_drop);
    return hip_result(hvc::branch_bwd_launch(a, (hipStream_t)stream), "branch_bwd");
}

int64_t hvc_colsum_workspace(int M, int N) {
    if (M < 1 || N < 1) return -1;
    return (int64_t)hvc::rowops_blocks(M) * N;
}

int hvc_colsum(const void* x, float* out, float* workspace, int M, int N, int dtype, void* stream) {
    if (!x || !out || !workspace) return fail(HVC_E_BADARG, "colsum: null operand");
    if (M < 1 || N < 1 || !dtype_ok(dtype)) return fail(HVC_E_BADARG, "colsum: bad shape/dtype");
    return hip_result(hvc::colsum_launch(x, workspace, out, M, N, hvc::rowops_blocks(M), dtype == HVC_BF16, (hipStream_t)stream), "colsum");
}

int hvc_cast(const void* x, void* y, int64_t n, int in_dtype, int out_dtype, void* stream) {
    if (!x || !y || n < 1) return fail(HVC_E_BADARG, "cast: bad operand");
    if (!dtype_ok(in_dtype) || !dtype_ok(out_dtype) || in_dtype == out_dtype) return fail(HVC_E_BADARG, "cast: bad dtype pair");
    return hip_result(hvc::cast_launch(x, y, n, in_dtype == HVC_BF16, out_dtype == HVC_BF16, (hipStream_t)stream), "cast");
}

static int fill_drr(hvc::DrrArgs& a, const void* vol, int B, int D, int H, int W, int axis, int exp_mode, float mu,
                    float out_scale, float clamp_min, int transpose_out, int dtype) {
    if (!vol) return fail(HVC_E_BADARG, "drr: null volume");
    if (B < 1 || D < 1 || H < 1 || W < 1) return fail(HVC_E_BADARG, "drr: empty dimension");
    if (axis != 0 && axis != 2) return fail(HVC_E_UNSUPPORTED, "drr: axis must be 0 (along D) or 2 (along W)");
    if (!dtype_ok(dtype)) return fail(HVC_E_BADARG, "drr: bad dtype");
    memset(&a, 0, sizeof(a));
    a.vol = vol; a.B = B; a.D = D; a.H = H; a.W = W; a.axis = axis; a.exp_mode = exp_mode != 0; a.mu = mu;
    a.out_scale = out_scale; a.clamp_min = clamp_min; a.transpose_out = (axis == 2) && transpose_out; a.is_bf16 = dtype == HVC_BF16;
    return 0;
}

int hvc_drr_fwd(const void* vol, void* out, int B, int D, int H, int W, int axis, int exp_mode, float mu,
                float out_scale, float clamp_min, int transpose_out, int dtype, void* stream) {
    hvc::DrrArgs a;
    int rc = fill_drr(a, vol, B, D, H, W, axis, exp_mode, mu, out_scale, clamp_min, transpose_out, dtype);
    if (rc) return rc;
    if (!out) return fail(HVC_E_BADARG, "drr_fwd: null output");
    a.out = out;
    return hip_result(hvc::drr_fwd_launch(a, (hipStream_t)stream), "drr_fwd");
}

int hvc_drr_bwd(const void* vol, const void* out, const void* dout, void* dvol, int B, int D, int H, int W, int axis,
                int exp_mode, float mu, float out_scale, float clamp_min, int transpose_out, int dtype, void* stream) {
    hvc::DrrArgs a;
    int rc = fill_drr(a, vol, B, D, H, W, axis, exp_mode, mu, out_scale, clamp_min, transpose_out, dtype);
    if (rc) return rc;
    if (!out || !dout || !dvol) return fail(HVC_E_BADARG, "drr_bwd: null operand");
    a.out = const_cast<void*>(out); a.dout = dout; a.dvol = dvol;
    return hip_result(hvc::drr_bwd_launch(a, (hipStream_t)stream), "drr_bwd");
}

static int fill_geom(hvc::ConvGeom& g, int B, int C, int SD, int SH, int SW, int KD, int KH, int KW, int stride,
                     int PD, int PH, int PW, int OD, int64_t Kp) {
    if (B < 1 || C < 1 || SD < 1 || SH < 1 || SW < 1 || KD < 1 || KH < 1 || KW < 1 || stride < 1 || PD < 0 || PH < 0 || PW < 0)
        return fail(HVC_E_BADARG, "conv geometry: bad extent");
    g.B = B; g.C = C; g.SD = SD; g.SH = SH; g.SW = SW; g.KD = KD; g.KH = KH; g.KW = KW; g.stride = stride;
    g.PD = PD; g.PH = PH; g.PW = PW;
    g.OD = OD > 0 ? OD : (SD + 2 * PD - KD) / stride + 1; g.OH = (SH + 2 * PH - KH) / stride + 1; g.OW = (SW + 2 * PW - KW) / stride + 1;
    if (g.OD < 1 || g.OH < 1 || g.OW < 1) return fail(HVC_E_BADARG, "conv geometry: empty output");
    g.M = (int64_t)B * g.OD * g.OH * g.OW;
    g.Kp = Kp;
    if (Kp < (int64_t)KD * KH * KW * C || Kp % 8) return fail(HVC_E_BADARG, "conv geometry: Kp must be a multiple of 8 >= taps*C");
    if (C % 8 == 0 && Kp != (int64_t)KD * KH * KW * C) return fail(HVC_E_BADARG, "conv geometry: Kp must equal taps*C when C % 8 == 0");
    return 0;
}

int hvc_im2col(const void* src, void* col, int B, int C, int SD, int SH, int SW, int KD, int KH, int KW, int stride,
               int PD, int PH, int PW, int OD, int64_t Kp, int dtype, void* stream) {
    hvc::ConvGeom g;
    int rc = fill_geom(g, B, C, SD, SH, SW, KD, KH, KW, stride, PD, PH, PW, OD, Kp);
    if (rc) return rc;
    if (!src || !col || !dtype_ok(dtype)) return fail(HVC_E_BADARG, "im2col: bad operand");
    if (!aligned16(col) || (C % 8 == 0 && !aligned16(src))) return fail(HVC_E_BADARG, "im2col: operands must be 16-byte aligned");
    return hip_result(hvc::im2col_launch(g, src, col, dtype == HVC_BF16, (hipStream_t)stream), "im2col");
}

int hvc_conv_c1_fwd(const void* x, const void* w2d, const float* bias, void* y, int B, int SD, int SH, int SW, int Cout, int stride,
                    void* stream) {
    if (!x || !w2d || !y) return fail(HVC_E_BADARG, "conv_c1_fwd: null operand");
    if (B < 1 || SD < 1 || SH < 1 || SW < 1) return fail(HVC_E_BADARG, "conv_c1_fwd: empty volume");
    if (!hvc::conv_c1_supported(Cout, stride)) return fail(HVC_E_UNSUPPORTED, "conv_c1_fwd: Cout must be 32 or 64, stride 1 or 2 (use im2col + gemm)");
    if (!aligned16(w2d) || !aligned16(y)) return fail(HVC_E_BADARG, "conv_c1_fwd: w2d and y must be 16-byte aligned");
    hvc::ConvC1Args a{};
    a.x = x; a.w2d = w2d; a.bias = bias; a.y = y; a.B = B; a.SD = SD; a.SH = SH; a.SW = SW; a.Cout = Cout; a.stride = stride;
    return hip_result(hvc::conv_c1_fwd_launch(a, (hipStream_t)stream), "conv_c1_fwd");
}

int64_t hvc_conv_c1_dw_workspace(int B, int SD, int SH, int SW, int Cout, int stride) {
    if (B < 1 || SD < 1 || SH < 1 || SW < 1 || !hvc::conv_c1_supported(Cout, stride)) return -1;
    return (int64_t)hvc::conv_c1_dw_parts(B, SD, SH, SW, stride) * Cout * 32;
}

int hvc_conv_c1_dw(const void* x, const void* dy, float* dw, float* workspace, int B, int SD, int SH, int SW, int Cout, int stride,
                   void* stream) {
    if (!x || !dy || !dw || !workspace) return fail(HVC_E_BADARG, "conv_c1_dw: null operand");
    if (B < 1 || SD < 1 || SH < 1 || SW < 1) return fail(HVC_E_BADARG, "conv_c1_dw: empty volume");
    if (!hvc::conv_c1_supported(Cout, stride)) return fail(HVC_E_UNSUPPORTED, "conv_c1_dw: Cout must be 32 or 64, stride 1 or 2 (use im2col + gemm)");
    if (!aligned16(dy)) return fail(HVC_E_BADARG, "conv_c1_dw: dy must be 16-byte aligned");
    hvc::ConvC1Args a{};
    a.x = x; a.dy = dy; a.workspace = workspace; a.B = B; a.SD = SD; a.SH = SH; a.SW = SW; a.Cout = Cout; a.stride = stride;
    return hip_result(hvc::conv_c1_dw_launch(a, dw, (hipStream_t)stream), "conv_c1_dw");
}

int hvc_conv_c1_dx(const void* dy, const void* wt, void* dx, int B, int D, int H, int W, int Cout, void* stream) {
    if (!dy || !wt || !dx) return fail(HVC_E_BADARG, "conv_c1_dx: null operand");
    if (B < 1 || D < 1 || H < 1 || W < 1) return fail(HVC_E_BADARG, "conv_c1_dx: empty volume");
    if (!hvc::conv_c1_supported(Cout, 1)) return fail(HVC_E_UNSUPPORTED, "conv_c1_dx: Cout must be 32 or 64 (use gemm + col2im)");
    if (!aligned16(dy) || !aligned16(wt)) return fail(HVC_E_BADARG, "conv_c1_dx: dy and wt must be 16-byte aligned");
    hvc::ConvC1Args a{};
    a.dy = dy; a.w2d = wt; a.y = dx; a.B = B; a.SD = D; a.SH = H; a.SW = W; a.Cout = Cout; a.stride = 1;
    return hip_result(hvc::conv_c1_dx_launch(a, (hipStream_t)stream), "conv_c1_dx");
}

int hvc_conv3_halo(const void* x, const void* wfrag, const float* bias, void* y, int B, int D, int H, int W, int CI, int CO, void* stream) {
    if (!x || !wfrag || !y) return fail(HVC_E_BADARG, "conv3_halo: null operand");
    if (B < 1 || D < 1 || H < 1 || W < 1) return fail(HVC_E_BADARG, "conv3_halo: empty volume");
    if (!hvc::conv3_halo_supported(CI, CO)) return fail(HVC_E_UNSUPPORTED, "conv3_halo: channel counts must be 32 or 64 (use conv_gemm)");
    if (!aligned16(x) || !aligned16(wfrag) || !aligned16(y)) return fail(HVC_E_BADARG, "conv3_halo: operands must be 16-byte aligned");
    hvc::Conv3Args a{};
    a.x = x; a.wfrag = wfrag; a.bias = bias; a.y = y; a.B = B; a.D = D; a.H = H; a.W = W; a.CI = CI; a.CO = CO;
    return hip_result(hvc::conv3_halo_launch(a, (hipStream_t)stream), "conv3_halo");
}

int hvc_conv_o1_fwd(const void* x, const void* w, const float* bias, void* y, int64_t M, int C, void* stream) {
    if (!x || !w || !y || M < 1) return fail(HVC_E_BADARG, "conv_o1_fwd: bad operand");
    if (!hvc::conv_o1_supported(C)) return fail(HVC_E_UNSUPPORTED, "conv_o1_fwd: C must be 8, 16, 32, 64 or 128 (use conv_gemm)");
    if (!aligned16(x) || !aligned16(w)) return fail(HVC_E_BADARG, "conv_o1_fwd: x and w must be 16-byte aligned");
    return hip_result(hvc::conv_o1_fwd_launch(x, w, bias, y, M, C, (hipStream_t)stream), "conv_o1_fwd");
}

int64_t hvc_conv_o1_bwd_workspace(int64_t M, int C) {
    if (M < 1 || !hvc::conv_o1_supported(C)) return -1;
    return (int64_t)hvc::conv_o1_bwd_blocks(M, C) * (C + 1);
}

int hvc_conv_o1_bwd(const void* x, const void* dy, const void* w, void* dx, float* dwb, float* workspace, int64_t M, int C, void* stream) {
    if (!x || !dy || !w || !dwb || !workspace || M < 1) return fail(HVC_E_BADARG, "conv_o1_bwd: bad operand");
    if (!hvc::conv_o1_supported(C)) return fail(HVC_E_UNSUPPORTED, "conv_o1_bwd: C must be 8, 16, 32, 64 or 128 (use gemm + col2im)");
    if (!aligned16(x) || !aligned16(w) || (dx && !aligned16(dx))) return fail(HVC_E_BADARG, "conv_o1_bwd: x, w and dx must be 16-byte aligned");
    return hip_result(hvc::conv_o1_bwd_launch(x, dy, w, dx, dwb, workspace, M, C, (hipStream_t)stream), "conv_o1_bwd");
}

int hvc_conv_gemm(int mode, const void* src, const void* other, void* out, int B, int C, int SD, int SH, int SW,
                  int KD, int KH, int KW, int stride, int PD, int PH, int PW, int flip, int N, int64_t ld_other, int64_t ld_out,
                  const float* bias, const float* residual, int64_t ldr, int residual_rows,
                  float* workspace, int64_t workspace_floats, int in_dtype, int out_dtype, void* stream) {
    hvc::ConvGeom cgeo;
    const int64_t K64 = (int64_t)KD * KH * KW * C;
    int rc = fill_geom(cgeo, B, C, SD, SH, SW, KD, KH, KW, stride, PD, PH, PW, 0, (K64 + 7) / 8 * 8);
    if (rc) return rc;
    if (mode != 0 && mode != 1) return fail(HVC_E_BADARG, "conv_gemm: mode must be 0 (patches x weights) or 1 (dy^T x patches)");
    if (!src || !other || !out || N < 1) return fail(HVC_E_BADARG, "conv_gemm: null operand");
    if (!dtype_ok(in_dtype) || !dtype_ok(out_dtype)) return fail(HVC_E_BADARG, "conv_gemm: bad dtype");
    if (in_dtype == HVC_F32 && out_dtype == HVC_BF16) return fail(HVC_E_UNSUPPORTED, "conv_gemm: f32 in / bf16 out not supported");
    if (C % 8) return fail(HVC_E_UNSUPPORTED, "conv_gemm: the implicit patch gather needs C % 8 == 0 (use im2col + gemm)");
    if (!aligned16(src)) return fail(HVC_E_BADARG, "conv_gemm: src must be 16-byte aligned");
    if (cgeo.M >= (1ll << 31) || K64 >= (1ll << 31)) return fail(HVC_E_UNSUPPORTED, "conv_gemm: more than 2^31 patch rows / columns");
    if (mode == 1 && (bias || residual)) return fail(HVC_E_BADARG, "conv_gemm: the weight-gradient form has no epilogue operands");
    if (mode == 1 && out_dtype != HVC_F32) return fail(HVC_E_BADARG, "conv_gemm: weight gradients are fp32");
    hvc::GemmArgs g;
    memset(&g, 0, sizeof(g));
    hvc::ConvGather& cg = g.cg;
    cg.src = src; cg.C = C; cg.SD = SD; cg.SH = SH; cg.SW = SW; cg.KD = KD; cg.KH = KH; cg.KW = KW; cg.stride = stride;
    cg.PD = PD; cg.PH = PH; cg.PW = PW; cg.flip = flip != 0; cg.M = cgeo.M; cg.K = (int)K64;
    const int64_t src_bytes = (int64_t)B * SD * SH * SW * C * (in_dtype == HVC_BF16 ? 2 : 4);
    const bool force64 = hvc::option(hvc::kOptConvForceAddr64) == 1;   // test hook for the >= 4 GiB path
    cg.bytes = (src_bytes < (1ll << 32) && !force64) ? (uint32_t)src_bytes : 0u;
    cg.dC = hvc::make_fastdiv((uint32_t)C); cg.dKW = hvc::make_fastdiv((uint32_t)KW); cg.dKH = hvc::make_fastdiv((uint32_t)KH);
    cg.dOW = hvc::make_fastdiv((uint32_t)cgeo.OW); cg.dOH = hvc::make_fastdiv((uint32_t)cgeo.OH); cg.dOD = hvc::make_fastdiv((uint32_t)cgeo.OD);
    g.C = out; g.ldc = ld_out; g.alpha = 1.f; g.act = 0;
    g.in_bf16 = in_dtype == HVC_BF16; g.out_bf16 = out_dtype == HVC_BF16;
    g.workspace = workspace; g.workspace_floats = workspace ? workspace_floats : 0;
    if (mode == 0) {            // out[M][N] = patches[M][K] . other[N][K]^T
        g.gather = cg.bytes ? 1 : 3;
        g.A = src; g.lda = K64; g.B = other; g.ldb = ld_other;
        g.M = (int)cgeo.M; g.N = N; g.K = (int)K64;
        g.bias = bias; g.residual = residual; g.ldr = ldr; g.residual_rows = residual_rows > 0 ? residual_rows : 0;
        g.rows_per_batch = g.M;
        g.vec_a = 1; g.vec_b = aligned16(other) && (ld_other % 8 == 0);
    } else {                    // out[N][K] = other[M][N]^T . patches[M][K]
        g.gather = cg.bytes ? 2 : 4;
        g.a_kmajor = g.b_kmajor = 1;
        g.A = other; g.lda = ld_other; g.B = src; g.ldb = K64;
        g.M = N; g.N = (int)K64; g.K = (int)cgeo.M;
        g.rows_per_batch = g.M;
        g.vec_a = aligned16(other) && (ld_other % 8 == 0); g.vec_b = 1;
    }
    g.vec_epi = (g.N % 8 == 0) && aligned16(out) && (ld_out % 8 == 0) && (!residual || (aligned16(residual) && ldr % 8 == 0)) &&
                (!workspace || aligned16(workspace));
    return hip_result(hvc::gemm_launch(g, (hipStream_t)stream), "conv_gemm");
}

// taps of one axis that reach input positions of parity class p (position = stride * i' + p), as a stride-1 window over dy:
// n taps, source coordinate i' + e0 + t for t = 0..n-1, tap t is kernel index k0 - t * stride
static void class_axis(int K, int stride, int P, int p, int& n, int& e0, int& k0) {
    const int kmin = (p + P) % stride;                    // smallest kernel index with (p + P - k) % stride == 0
    n = kmin < K ? (K - 1 - kmin) / stride + 1 : 0;
    k0 = kmin + (n - 1) * stride;                         // largest such index <-> smallest source offset
    e0 = n ? (p + P - k0) / stride : 0;                   // may be negative (reads in front of dy: zero) or positive
}

int hvc_conv_dx_class_columns(int KD, int KH, int KW, int stride, int PD, int PH, int PW, int cd, int ch, int cw, int* first_tap, int* ntaps) {
    if (KD < 1 || KH < 1 || KW < 1 || stride < 1 || cd < 0 || ch < 0 || cw < 0 || cd >= stride || ch >= stride || cw >= stride)
        return fail(HVC_E_BADARG, "conv_dx_class_columns: bad geometry");
    int off = 0;
    for (int d = 0; d < stride; ++d)
        for (int h = 0; h < stride; ++h)
            for (int w = 0; w < stride; ++w) {
                int nd, nh, nw, e, k;
                class_axis(KD, stride, PD, d, nd, e, k);
                class_axis(KH, stride, PH, h, nh, e, k);
                class_axis(KW, stride, PW, w, nw, e, k);
                if (d == cd && h == ch && w == cw) {
                    if (first_tap) *first_tap = off;
                    if (ntaps) *ntaps = nd * nh * nw;
                    return 0;
                }
                off += nd * nh * nw;
            }
    return fail(HVC_E_BADARG, "conv_dx_class_columns: class not found");
}

int hvc_conv_dx_class(const void* dy, const void* wclass, void* dx, int B, int Cout, int OD, int OH, int OW,
                      int Cin, int SD, int SH, int SW, int KD, int KH, int KW, int stride, int PD, int PH, int PW,
                      int cd, int ch, int cw, int64_t ld_w, int dtype, void* stream) {
    if (!dy || !wclass || !dx) return fail(HVC_E_BADARG, "conv_dx_class: null operand");
    if (B < 1 || Cout < 1 || Cin < 1 || OD < 1 || OH < 1 || OW < 1 || SD < 1 || SH < 1 || SW < 1 || KD < 1 || KH < 1 || KW < 1 || stride < 1)
        return fail(HVC_E_BADARG, "conv_dx_class: bad extent");
    if (cd < 0 || ch < 0 || cw < 0 || cd >= stride || ch >= stride || cw >= stride) return fail(HVC_E_BADARG, "conv_dx_class: class out of range");
    if (!dtype_ok(dtype)) return fail(HVC_E_BADARG, "conv_dx_class: bad dtype");
    if (Cout % 8 || Cin % 8) return fail(HVC_E_UNSUPPORTED, "conv_dx_class: channel counts must be multiples of 8");
    if (!aligned16(dy) || !aligned16(wclass) || !aligned16(dx) || ld_w % 8) return fail(HVC_E_BADARG, "conv_dx_class: operands must be 16-byte aligned");
    int nd, nh, nw, ed, eh, ew, kd0, kh0, kw0;
    class_axis(KD, stride, PD, cd, nd, ed, kd0);
    class_axis(KH, stride, PH, ch, nh, eh, kh0);
    class_axis(KW, stride, PW, cw, nw, ew, kw0);
    const int cD = cd < SD ? (SD - cd + stride - 1) / stride : 0, cH = ch < SH ? (SH - ch + stride - 1) / stride : 0,
              cW = cw < SW ? (SW - cw + stride - 1) / stride : 0;                  // input positions of this class per axis
    if (nd * nh * nw == 0 || cD * cH * cW == 0) return fail(HVC_E_BADARG, "conv_dx_class: empty class (no tap reaches it: those dx values are zero)");
    const int64_t M = (int64_t)B * cD * cH * cW, K64 = (int64_t)nd * nh * nw * Cout;
    if (M >= (1ll << 31) || K64 >= (1ll << 31)) return fail(HVC_E_UNSUPPORTED, "conv_dx_class: more than 2^31 rows / columns");
    hvc::GemmArgs g;
    memset(&g, 0, sizeof(g));
    hvc::ConvGather& cg = g.cg;
    // a stride-1 window of (nd, nh, nw) taps over dy: row (b, i'd, i'h, i'w) reads dy at i' + e0 + t, i.e. "padding" -e0
    cg.src = dy; cg.C = Cout; cg.SD = OD; cg.SH = OH; cg.SW = OW; cg.KD = nd; cg.KH = nh; cg.KW = nw; cg.stride = 1;
    cg.PD = -ed; cg.PH = -eh; cg.PW = -ew; cg.flip = 0; cg.M = M; cg.K = (int)K64;
    const int64_t src_bytes = (int64_t)B * OD * OH * OW * Cout * (dtype == HVC_BF16 ? 2 : 4);
    const bool force64 = hvc::option(hvc::kOptConvForceAddr64) == 1;
    cg.bytes = (src_bytes < (1ll << 32) && !force64) ? (uint32_t)src_bytes : 0u;
    cg.dC = hvc::make_fastdiv((uint32_t)Cout); cg.dKW = hvc::make_fastdiv((uint32_t)nw); cg.dKH = hvc::make_fastdiv((uint32_t)nh);
    cg.dOW = hvc::make_fastdiv((uint32_t)cW); cg.dOH = hvc::make_fastdiv((uint32_t)cH); cg.dOD = hvc::make_fastdiv((uint32_t)cD);
    g.gather = cg.bytes ? 1 : 3;
    g.A = dy; g.lda = K64; g.B = wclass; g.ldb = ld_w;
    g.M = (int)M; g.N = Cin; g.K = (int)K64; g.rows_per_batch = g.M;
    g.C = dx; g.ldc = Cin; g.alpha = 1.f; g.act = 0;
    g.in_bf16 = g.out_bf16 = dtype == HVC_BF16;
    g.vec_a = 1; g.vec_b = 1; g.vec_epi = 1;
    // rows of the class result land on the interleaved positions (stride i' + class) of dx [B][SD][SH][SW][Cin]
    g.omap = 1;
    g.oW = hvc::make_fastdiv((uint32_t)cW); g.oH = hvc::make_fastdiv((uint32_t)cH); g.oD = hvc::make_fastdiv((uint32_t)cD);
    g.o_sw = (int64_t)stride * Cin; g.o_sh = (int64_t)stride * SW * Cin; g.o_sd = (int64_t)stride * SH * SW * Cin;
    g.o_sb = (int64_t)SD * SH * SW * Cin;
    g.o_base = (((int64_t)cd * SH + ch) * SW + cw) * Cin;
    return hip_result(hvc::gemm_launch(g, (hipStream_t)stream), "conv_dx_class");
}

int hvc_col2im(const void* dcol, void* dsrc, int B, int C, int SD, int SH, int SW, int KD, int KH, int KW, int stride,
               int PD, int PH, int PW, int OD, int64_t Kp, int dtype, void* stream) {
    hvc::ConvGeom g;
    int rc = fill_geom(g, B, C, SD, SH, SW, KD, KH, KW, stride, PD, PH, PW, OD, Kp);
    if (rc) return rc;
    if (!dcol || !dsrc || !dtype_ok(dtype)) return fail(HVC_E_BADARG, "col2im: bad operand");
    if (C % 8 == 0 && !(aligned16(dcol) && aligned16(dsrc))) return fail(HVC_E_BADARG, "col2im: operands must be 16-byte aligned");
    return hip_result(hvc::col2im_launch(g, dcol, dsrc, dtype == HVC_BF16, (hipStream_t)stream), "col2im");
}

int hvc_trilinear_fwd(const float* src, float* dst, int B, int d, int h, int w, int D, int H, int W, int align_corners, void* stream) {
    if (!src || !dst || B < 1 || d < 1 || h < 1 || w < 1 || D < 1 || H < 1 || W < 1) return fail(HVC_E_BADARG, "trilinear: bad operand");
    return hip_result(hvc::trilinear_launch(src, dst, B, d, h, w, D, H, W, align_corners != 0, false, (hipStream_t)stream), "trilinear_fwd");
}
int64_t hvc_trilinear_bwd_workspace(int B, int d, int h, int w, int D, int H, int W) {
    if (B < 1 || d < 1 || h < 1 || w < 1 || D < 1 || H < 1 || W < 1) return 0;
    return hvc::trilinear_bwd_workspace_floats(B, d, h, w, D, H, W);
}
int hvc_trilinear_bwd(const float* dout, float* dsrc, float* workspace, int B, int d, int h, int w, int D, int H, int W, int align_corners,
                      void* stream) {
    if (!dout || !dsrc || B < 1 || d < 1 || h < 1 || w < 1 || D < 1 || H < 1 || W < 1) return fail(HVC_E_BADARG, "trilinear: bad operand");
    if (workspace)
        return hip_result(hvc::trilinear_bwd_separable_launch(dout, dsrc, workspace, B, d, h, w, D, H, W, align_corners != 0, (hipStream_t)stream),
                          "trilinear_bwd");
    return hip_result(hvc::trilinear_launch(dout, dsrc, B, d, h, w, D, H, W, align_corners != 0, true, (hipStream_t)stream), "trilinear_bwd");
}

static bool norm_c_ok(int C) { return C >= 8 && C <= 512 && C % 8 == 0 && 256 % (C / 8) == 0; }

int64_t hvc_norm_workspace(int B, int P, int C, int G) {
    if (B < 1 || P < 1 || C < 1 || G < 1) return -1;
    return (int64_t)B * hvc::norm_chunks(P) * 2 * C + 2 * (int64_t)B * G + 2 * (int64_t)C;
}

int hvc_groupnorm_act_fwd(const void* x, void* y, const float* gamma, const float* beta, float* stats, float* workspace,
                          int B, int P, int C, int G, float eps, int act, int dtype, void* stream) {
    if (!x || !y || !gamma || !beta || !stats || !workspace || !dtype_ok(dtype)) return fail(HVC_E_BADARG, "groupnorm: bad operand");
    if (B < 1 || P < 1 || G < 1 || C % G) return fail(HVC_E_BADARG, "groupnorm: bad shape");
    if (!norm_c_ok(C)) return fail(HVC_E_UNSUPPORTED, "groupnorm: C must be 8,16,32,64,128,256 or 512");
    hvc::NormArgs a; memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.gamma = gamma; a.beta = beta; a.stats = stats; a.partial = workspace;
    if (act != 0 && act != 1) return fail(HVC_E_BADARG, "groupnorm: act must be 0 (SiLU) or 1 (GELU)");
    a.B = B; a.P = P; a.C = C; a.G = G; a.eps = eps; a.is_bf16 = dtype == HVC_BF16; a.act = act;
    return hip_result(hvc::groupnorm_silu_fwd_launch(a, (hipStream_t)stream), "groupnorm_silu_fwd");
}

int hvc_groupnorm_act_bwd(const void* x, const void* dy, void* dx, const float* gamma, const float* beta, const float* stats,
                          float* dgamma, float* dbeta, float* workspace, int B, int P, int C, int G, int act, int dtype, void* stream) {
    if (!x || !dy || !dx || !gamma || !beta || !stats || !dgamma || !dbeta || !workspace || !dtype_ok(dtype))
        return fail(HVC_E_BADARG, "groupnorm_bwd: bad operand");
    if (B < 1 || P < 1 || G < 1 || C % G) return fail(HVC_E_BADARG, "groupnorm: bad shape");
    if (!norm_c_ok(C)) return fail(HVC_E_UNSUPPORTED, "groupnorm: C must be 8,16,32,64,128,256 or 512");
    hvc::NormArgs a; memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy; a.dx = dx; a.gamma = gamma; a.beta = beta; a.stats = const_cast<float*>(stats);
    a.dgamma = dgamma; a.dbeta = dbeta; a.partial = workspace;
    a.gsum = workspace + (int64_t)B * hvc::norm_chunks(P) * 2 * C;
    if (act != 0 && act != 1) return fail(HVC_E_BADARG, "groupnorm: act must be 0 (SiLU) or 1 (GELU)");
    a.B = B; a.P = P; a.C = C; a.G = G; a.is_bf16 = dtype == HVC_BF16; a.act = act;
    return hip_result(hvc::groupnorm_silu_bwd_launch(a, (hipStream_t)stream), "groupnorm_silu_bwd");
}

static int fill_pool(hvc::PoolGeom& pg, int N, int H, int W, int C, int k, int s, int p) {
    if (N < 1 || H < 1 || W < 1 || k < 1 || s < 1 || p < 0 || k > 15) return fail(HVC_E_BADARG, "bn_relu_pool: bad geometry");
    if (!norm_c_ok(C)) return fail(HVC_E_UNSUPPORTED, "bn_relu_pool: C must be 8,16,32,64,128,256 or 512");
    pg.N = N; pg.H = H; pg.W = W; pg.C = C; pg.k = k; pg.s = s; pg.p = p;
    pg.HP = (H + 2 * p - k) / s + 1; pg.WP = (W + 2 * p - k) / s + 1;
    if (pg.HP < 1 || pg.WP < 1) return fail(HVC_E_BADARG, "bn_relu_pool: empty output");
    return 0;
}

int hvc_bn_relu_pool_fwd(const void* x, void* y, uint8_t* amax, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float* stats, float* workspace, int N, int H, int W, int C, int k, int s, int p,
                         int training, float eps, float momentum, int dtype, void* stream) {
    hvc::PoolGeom pg;
    int rc = fill_pool(pg, N, H, W, C, k, s, p);
    if (rc) return rc;
    if (!x || !y || !gamma || !beta || !stats || !workspace || !dtype_ok(dtype)) return fail(HVC_E_BADARG, "bn_relu_pool: bad operand");
    if (!training && (!running_mean || !running_var)) return fail(HVC_E_BADARG, "bn_relu_pool: eval mode needs running stats");
    if (k > 1 && !amax) return fail(HVC_E_BADARG, "bn_relu_pool: pooling needs the arg-max buffer");
    hvc::NormArgs a; memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.amax = k > 1 ? amax : nullptr; a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var;
    a.stats = stats; a.partial = workspace; a.training = training != 0; a.eps = eps; a.momentum = momentum; a.is_bf16 = dtype == HVC_BF16;
    return hip_result(hvc::bn_relu_pool_fwd_launch(a, pg, (hipStream_t)stream), "bn_relu_pool_fwd");
}

int hvc_bn_relu_pool_bwd(const void* x, const void* dy, const uint8_t* amax, void* dx, const float* gamma, const float* beta,
                         const float* stats, float* dgamma, float* dbeta, float* workspace, int N, int H, int W, int C,
                         int k, int s, int p, int training, int dtype, void* stream) {
    hvc::PoolGeom pg;
    int rc = fill_pool(pg, N, H, W, C, k, s, p);
    if (rc) return rc;
    if (!x || !dy || !dx || !gamma || !beta || !stats || !dgamma || !dbeta || !workspace || !dtype_ok(dtype))
        return fail(HVC_E_BADARG, "bn_relu_pool_bwd: bad operand");
    if (k > 1 && !amax) return fail(HVC_E_BADARG, "bn_relu_pool_bwd: pooling needs the arg-max buffer");
    hvc::NormArgs a; memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy; a.dx = dx; a.amax = k > 1 ? const_cast<uint8_t*>(amax) : nullptr; a.gamma = gamma; a.beta = beta;
    a.stats = const_cast<float*>(stats); a.dgamma = dgamma; a.dbeta = dbeta; a.partial = workspace;
    a.gsum = workspace + (int64_t)N * hvc::norm_chunks(H * W) * 2 * C;
    a.training = training != 0; a.is_bf16 = dtype == HVC_BF16;
    return hip_result(hvc::bn_relu_pool_bwd_launch(a, pg, (hipStream_t)stream), "bn_relu_pool_bwd");
}

int64_t hvc_ssim_l1_workspace(int B, int D, int H, int W) {
    if (B < 1 || D < 1 || H < 1 || W < 1) return -1;
    const int64_t nvox = (int64_t)B * D * H * W;
    return 10 * nvox + 2 * (int64_t)hvc::loss_blocks(nvox);
}

static int fill_loss(hvc::LossArgs& a, const float* pred, const float* target, int B, int D, int H, int W, int window, float l1_w, float ssim_w) {
    if (!pred || !target || B < 1 || D < 1 || H < 1 || W < 1 || window < 1 || !(window & 1)) return fail(HVC_E_BADARG, "ssim_l1: bad operand");
    memset(&a, 0, sizeof(a));
    a.pred = pred; a.target = target; a.B = B; a.D = D; a.H = H; a.W = W; a.window = window; a.l1_w = l1_w; a.ssim_w = ssim_w;
    return 0;
}

int hvc_ssim_l1_fwd(const float* pred, const float* target, float* out3, float* gmaps, float* workspace,
                    int B, int D, int H, int W, int window, float l1_w, float ssim_w, void* stream) {
    hvc::LossArgs a;
    int rc = fill_loss(a, pred, target, B, D, H, W, window, l1_w, ssim_w);
    if (rc) return rc;
    if (!out3 || !gmaps || !workspace) return fail(HVC_E_BADARG, "ssim_l1_fwd: null output");
    a.out = out3; a.gmaps = gmaps; a.workspace = workspace;
    return hip_result(hvc::ssim_l1_fwd_launch(a, (hipStream_t)stream), "ssim_l1_fwd");
}

int hvc_ssim_l1_bwd(const float* pred, const float* target, const float* gmaps, const float* gscale, float* dpred, float* workspace,
                    int B, int D, int H, int W, int window, float l1_w, float ssim_w, void* stream) {
    hvc::LossArgs a;
    int rc = fill_loss(a, pred, target, B, D, H, W, window, l1_w, ssim_w);
    if (rc) return rc;
    if (!gmaps || !dpred || !workspace) return fail(HVC_E_BADARG, "ssim_l1_bwd: null operand");
    a.gmaps = const_cast<float*>(gmaps); a.gscale = gscale; a.dpred = dpred; a.workspace = workspace;
    return hip_result(hvc::ssim_l1_bwd_launch(a, (hipStream_t)stream), "ssim_l1_bwd");
}

int64_t hvc_tv3d_workspace(int B, int D, int H, int W) {
    if (B < 1 || D < 1 || H < 1 || W < 1) return -1;
    return 3 * (int64_t)hvc::loss_blocks((int64_t)B * D * H * W);
}

int hvc_tv3d_fwd(const float* vol, float* means3, float* workspace, int B, int D, int H, int W, float eps, void* stream) {
    if (!vol || !means3 || !workspace || B < 1 || D < 1 || H < 1 || W < 1 || !(eps >= 0.f)) return fail(HVC_E_BADARG, "tv3d_fwd: bad operand");
    hvc::TvArgs a;
    memset(&a, 0, sizeof(a));
    a.vol = vol; a.out = means3; a.workspace = workspace; a.B = B; a.D = D; a.H = H; a.W = W; a.eps = eps;
    return hip_result(hvc::tv3d_fwd_launch(a, (hipStream_t)stream), "tv3d_fwd");
}

int hvc_tv3d_bwd(const float* vol, const float* gscale, float* dvol, int B, int D, int H, int W, float eps, void* stream) {
    if (!vol || !gscale || !dvol || B < 1 || D < 1 || H < 1 || W < 1 || !(eps >= 0.f)) return fail(HVC_E_BADARG, "tv3d_bwd: bad operand");
    hvc::TvArgs a;
    memset(&a, 0, sizeof(a));
    a.vol = vol; a.gscale = gscale; a.dvol = dvol; a.B = B; a.D = D; a.H = H; a.W = W; a.eps = eps;
    return hip_result(hvc::tv3d_bwd_launch(a, (hipStream_t)stream), "tv3d_bwd");
}

int64_t hvc_spectral_l1_workspace(int B, int D, int H, int W) {
    if (B < 1 || D < 1 || H < 1 || W < 1) return -1;
    return 2 * (int64_t)hvc::loss_blocks((int64_t)B * D * H * W);
}

int hvc_spectral_l1_fwd(const float* pred_spec, const float* target_spec, float* out2, float* workspace, int B, int D, int H, int W, void* stream) {
    if (!pred_spec || !target_spec || !out2 || !workspace || B < 1 || D < 1 || H < 1 || W < 1) return fail(HVC_E_BADARG, "spectral_l1_fwd: bad operand");
    hvc::SpecArgs a;
    memset(&a, 0, sizeof(a));
    a.pred = pred_spec; a.target = target_spec; a.out = out2; a.workspace = workspace; a.B = B; a.D = D; a.H = H; a.W = W;
    return hip_result(hvc::spec_l1_fwd_launch(a, (hipStream_t)stream), "spectral_l1_fwd");
}

int hvc_spectral_l1_bwd(const float* pred_spec, const float* target_spec, const float* gscale, float* dpred_spec, int B, int D, int H, int W,
                        void* stream) {
    if (!pred_spec || !target_spec || !gscale || !dpred_spec || B < 1 || D < 1 || H < 1 || W < 1) return fail(HVC_E_BADARG, "spectral_l1_bwd: bad operand");
    hvc::SpecArgs a;
    memset(&a, 0, sizeof(a));
    a.pred = pred_spec; a.target = target_spec; a.gscale = gscale; a.dpred = dpred_spec; a.B = B; a.D = D; a.H = H; a.W = W;
    return hip_result(hvc::spec_l1_bwd_launch(a, (hipStream_t)stream), "spectral_l1_bwd");
}

int64_t hvc_resize_loss_workspace(int B, int S1, int S2) {
    if (B < 1 || S1 < 1 || S2 < 1) return -1;
    return hvc::resize_loss_blocks(B, S1, S2);
}

static int fill_resize_loss(hvc::ResizeLossArgs& a, const float* proj, const float* target, int B, int h, int w, int S1, int S2, int64_t tb,
                            int align_corners, int mode) {
    if (!proj || !target) return fail(HVC_E_BADARG, "resize_loss: null operand");
    if (B < 1 || h < 1 || w < 1 || S1 < 1 || S2 < 1 || tb < (int64_t)S1 * S2) return fail(HVC_E_BADARG, "resize_loss: bad geometry");
    if (mode < 0 || mode > 1) return fail(HVC_E_BADARG, "resize_loss: mode is 0 (L1) or 1 (MSE)");
    memset(&a, 0, sizeof(a));
    a.proj = proj; a.target = target; a.B = B; a.h = h; a.w = w; a.S1 = S1; a.S2 = S2; a.target_bstride = tb;
    a.align_corners = align_corners ? 1 : 0; a.mode = mode;
    return 0;
}

int hvc_resize_loss_fwd(const float* proj, const float* target, float* out1, float* workspace, int B, int h, int w, int S1, int S2,
                        int64_t target_bstride, int align_corners, int mode, void* stream) {
    hvc::ResizeLossArgs a;
    int rc = fill_resize_loss(a, proj, target, B, h, w, S1, S2, target_bstride, align_corners, mode);
    if (rc) return rc;
    if (!out1 || !workspace) return fail(HVC_E_BADARG, "resize_loss_fwd: null output");
    a.out = out1; a.workspace = workspace;
    return hip_result(hvc::resize_loss_fwd_launch(a, (hipStream_t)stream), "resize_loss_fwd");
}

int hvc_resize_loss_grad(const float* proj, const float* target, const float* gscale, float* dresized, int B, int h, int w, int S1, int S2,
                         int64_t target_bstride, int align_corners, int mode, void* stream) {
    hvc::ResizeLossArgs a;
    int rc = fill_resize_loss(a, proj, target, B, h, w, S1, S2, target_bstride, align_corners, mode);
    if (rc) return rc;
    if (!gscale || !dresized) return fail(HVC_E_BADARG, "resize_loss_grad: null operand");
    a.gscale = gscale; a.dres = dresized;
    return hip_result(hvc::resize_loss_grad_launch(a, (hipStream_t)stream), "resize_loss_grad");
}

int64_t hvc_view_mean_gap_workspace(int B, int P, int E) {
    if (B < 1 || P < 1 || E < 1) return -1;
    return (int64_t)B * hvc::view_gap_chunks(P) * E;
}

static int fill_view_gap(hvc::ViewGapArgs& a, int B, int V, int P, int E, int dtype) {
    if (B < 1 || V < 1 || P < 1 || E < 8 || (E % 8) != 0 || E > 2048 || (256 % (E / 8)) != 0)
        return fail(HVC_E_BADARG, "view_mean_gap: bad geometry (E / 8 must divide 256)");
    if (!dtype_ok(dtype)) return fail(HVC_E_BADARG, "view_mean_gap: bad dtype");
    memset(&a, 0, sizeof(a));
    a.B = B; a.V = V; a.P = P; a.E = E; a.is_bf16 = dtype == HVC_BF16;
    return 0;
}

int hvc_view_mean_gap_fwd(const void* feats, float* mean, float* pooled, float* workspace, int B, int V, int P, int E, int dtype, void* stream) {
    hvc::ViewGapArgs a;
    int rc = fill_view_gap(a, B, V, P, E, dtype);
    if (rc) return rc;
    if (!feats || !mean || !pooled || !workspace || !aligned16(feats) || !aligned16(mean)) return fail(HVC_E_BADARG, "view_mean_gap_fwd: null / unaligned operand");
    a.f = feats; a.mean = mean; a.pooled = pooled; a.workspace = workspace;
    return hip_result(hvc::view_mean_gap_fwd_launch(a, (hipStream_t)stream), "view_mean_gap_fwd");
}

int hvc_view_mean_gap_bwd(const float* dmean, const float* dpooled, void* dfeats, int B, int V, int P, int E, int dtype, void* stream) {
    hvc::ViewGapArgs a;
    int rc = fill_view_gap(a, B, V, P, E, dtype);
    if (rc) return rc;
    if (!dfeats || (!dmean && !dpooled) || !aligned16(dfeats)) return fail(HVC_E_BADARG, "view_mean_gap_bwd: null / unaligned operand");
    a.dmean = dmean; a.dpooled = dpooled; a.df = dfeats;
    return hip_result(hvc::view_mean_gap_bwd_launch(a, (hipStream_t)stream), "view_mean_gap_bwd");
}

}  // extern "C"
