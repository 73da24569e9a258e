// Whole-iteration driver for the BinaryMFPenalty / WNMF multiplicative-update loop
// (PyBMF/models/BinaryMFPenalty.py:61-115, PyBMF/models/WNMF.py:51-89) and kernel timing for bench.py.
//
// Everything is enqueued on one stream; nothing returns to the host inside an iteration.  Early stopping
// (PyBMF/models/BaseModelTools.py:299-343) is evaluated on the device by the finalize kernel, which raises a
// device flag; every later kernel starts by reading that flag and exits, so the host can enqueue max_iter+1
// iterations without a round trip and still end with exactly the reference's factors and log rows.
#include "common.h"
#include "comm.h"

#include <cstdlib>
#include <vector>

int bmf_xf_bits_launch(const uint32_t* Abits, int64_t rows_pad, int64_t ldw, int64_t red_words, const uint16_t* panel,
                        int64_t ldp, int terms, int kp, float* out, int64_t slab_stride, int splits, int panel_kind,
                        const float* colscale, const int32_t* stop, hipStream_t s);
int bmf_mae_launch(const uint32_t* XTbits, int64_t ldxt, int64_t m_pad, int64_t n_pad, const float* U, const float* V, int kp,
                   uint16_t* ws, double* sum, const int32_t* stop, hipStream_t s, int one_product, int x_tiled, int zero_sums);
int bmf_panel_f16_launch(const float* F, int64_t rows_pad, int64_t ldf, int kp, uint16_t* panel, int64_t ldp, float* ws,
                         float* scale, bool have_blockmax, const int32_t* stop, hipStream_t s);
int bmf_panel_i8_launch(const double* F64, const float* F, int64_t rows_pad, int64_t ldf, int kp, int limbs, int8_t* panel, int64_t ldp,
                        float* ws, float* scale, bool have_blockmax, const int32_t* stop, hipStream_t s, bool have_scale,
                        const float* flags, float* colscale_out, const float* rslabs, int rcount, int rn, float* rout32, double* rout64);
int bmf_gram_partial_launch(const float* F, int64_t rows_pad, int64_t ldf, int kp, float* slabs, int blocks, const float* blockmax, int limbs,
                            float* scale, const int32_t* stop, hipStream_t s, int fused);
int bmf_xf_bits_i8_launch(const uint32_t* Abits, int64_t rows_pad, int64_t ldw, int64_t red_words, const int8_t* panel, int64_t ldp,
                          int limbs, const float* colscale, int kp, int col0, int ncols, float* out, int64_t slab_stride, int splits,
                          int a_tiled, const int32_t* stop, hipStream_t s);
int bmf_cover_launch(const uint32_t* Xbits, int64_t rows_pad, int64_t ldx, int64_t words, const uint64_t* rowbits,
                     const uint32_t* colbits, int64_t ldcb, int kp, unsigned long long* counts, const int32_t* stop,
                     hipStream_t s);
int bmf_residual_launch(const uint32_t* Xbits, int64_t m_pad, int64_t ldx, int m, int n, const float* A, const float* B,
                        const float* dA, const float* dB, int kp, double* sums, const int32_t* stop, hipStream_t s);

// ---------------------------------------------------------------------------------------------------
// kernel timing: hipEvent pairs around the bits-GEMM launches of the driver
// ---------------------------------------------------------------------------------------------------
namespace {
struct Timer {
    bool on = false, open = false;
    int cap = 0, used = 0, every = 1, seen = 0;
    std::vector<hipEvent_t> ev;  // 2 per launch
} g_timer;
}  // namespace

// every `every`-th launch is bracketed (an event pair costs ~6 us of stream time around a 340 us kernel)
void bmf_timer_begin(hipStream_t s) {
    g_timer.open = g_timer.on && g_timer.used < g_timer.cap && (g_timer.seen++ % g_timer.every) == 0;
    if (g_timer.open) (void)hipEventRecord(g_timer.ev[2 * g_timer.used], s);
}
void bmf_timer_end(hipStream_t s) {
    if (g_timer.open) {
        (void)hipEventRecord(g_timer.ev[2 * g_timer.used + 1], s);
        ++g_timer.used;
        g_timer.open = false;
    }
}

extern "C" int bmf_timer_enable(int max_launches) {
    BMF_REQUIRE(max_launches >= 1 && max_launches <= (1 << 20), "bmf_timer_enable: max_launches out of range");
    bmf_timer_disable();
    g_timer.ev.resize(2 * (size_t)max_launches);
    for (auto& e : g_timer.ev) BMF_HIP_CHECK(hipEventCreate(&e));
    g_timer.cap = max_launches;
    g_timer.used = 0;
    g_timer.seen = 0;
    g_timer.every = 1;
    g_timer.on = true;
    return BMF_OK;
}

extern "C" int bmf_timer_stride(int every) {
    BMF_REQUIRE(every >= 1, "bmf_timer_stride: every must be >= 1");
    g_timer.every = every;
    g_timer.seen = 0;
    return BMF_OK;
}

extern "C" int bmf_timer_read(int* launches, double* total_ms) {
    BMF_REQUIRE(launches && total_ms, "bmf_timer_read: null pointer");
    double tot = 0.0;
    for (int i = 0; i < g_timer.used; ++i) {
        BMF_HIP_CHECK(hipEventSynchronize(g_timer.ev[2 * i + 1]));
        float ms = 0.f;
        BMF_HIP_CHECK(hipEventElapsedTime(&ms, g_timer.ev[2 * i], g_timer.ev[2 * i + 1]));
        tot += ms;
    }
    *launches = g_timer.used;
    *total_ms = tot;
    g_timer.used = 0;
    return BMF_OK;
}

extern "C" int bmf_timer_disable(void) {
    for (auto& e : g_timer.ev) (void)hipEventDestroy(e);
    g_timer.ev.clear();
    g_timer.on = false;
    g_timer.cap = g_timer.used = 0;
    return BMF_OK;
}

// ---------------------------------------------------------------------------------------------------
// scalar plumbing kernels
// ---------------------------------------------------------------------------------------------------
namespace {

// local partial sums -> comm block (what gets all-reduced); resets the local integer counters.  Called by all threads of a block
// of >= 256 threads; `sh`: 768 doubles of LDS.
__device__ __forceinline__ void gather_body(const double* __restrict__ partU, int nbU, const double* __restrict__ partV, int nbV,
                                            unsigned long long* __restrict__ counts, double* __restrict__ comm,
                                            double* __restrict__ scal, double* sh) {
    const int tid = threadIdx.x;
    if (tid < 256) {
        double regU = 0.0, dot = 0.0, regV = 0.0;
        for (int b = tid; b < nbU; b += 256) {
            regU += partU[2 * b];
            dot += partU[2 * b + 1];
        }
        for (int b = tid; b < nbV; b += 256) regV += partV[2 * b];
        sh[tid] = regU;
        sh[256 + tid] = dot;
        sh[512 + tid] = regV;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            sh[tid] += sh[tid + o];
            sh[256 + tid] += sh[256 + tid + o];
            sh[512 + tid] += sh[512 + tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        comm[0] = sh[256];  // sum U o (X V)
        comm[1] = sh[0];    // sum (U^2 - U)^2
        comm[2] = (double)counts[0];
        comm[3] = (double)counts[1];
        scal[0] = sh[512];  // sum (V^2 - V)^2 (V is replicated: not all-reduced)
        counts[0] = 0ull;
        counts[1] = 0ull;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void gather_kernel(const double* __restrict__ partU, int nbU,
                                                      const double* __restrict__ partV, int nbV,
                                                      unsigned long long* __restrict__ counts,
                                                      double* __restrict__ comm, double* __restrict__ scal,
                                                      const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) {   // after the stop nothing rebuilds the exchange block, but a row-sharded run keeps all-reducing it until the
        if (threadIdx.x < 8) comm[threadIdx.x] = 0.0;   // host notices: zeros stay zeros (the values of the last row are in the log)
        return;
    }
    __shared__ double sh[768];
    gather_body(partU, nbU, partV, nbV, counts, comm, scal, sh);
}

__global__ __launch_bounds__(256) void zero_mae_kernel(double* comm, const int32_t* stop) {
    if (stop && *stop != 0) return;
    if (threadIdx.x < 2) comm[4 + threadIdx.x] = 0.0;
}

// comm (all-reduced) -> log row, fp32 U^T U for the next V update, early-stop flag.  do_gather: the single-GPU loop has nothing to
// exchange between the gather and this kernel, so the gather runs here (one launch of ~5 us less per iteration).
__device__ __forceinline__ void finalize_body(const bmf_penalty_state& st, int iter, double reg_used, int max_iter, int do_gather, double* sh) {
    const int sflag = *st.stop;
    if (sflag != 0 && iter > sflag) return;  // rows after the stop iteration do not exist in the reference
    if (do_gather && sflag == 0)
        gather_body(st.partU, (int)(st.m_pad / 128), st.partV, (int)(st.n_pad / 128), st.counts, st.comm, st.scal, sh);
    const int kk = st.kp * st.kp;
    const double* GU = st.comm + 8;
    double b = 0.0;
    for (int i = threadIdx.x; i < kk; i += 1024) {  // <= 4 independent loads per thread
        b += GU[i] * st.GV64[i];
        st.GU[i] = (float)GU[i];
    }
    sh[threadIdx.x] = b;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    b = sh[0];
    const double a = st.comm[0];
    const double rec = 0.5 * (st.sum_x - 2.0 * a + b);  // 0.5 ||X - U V^T||^2 = 0.5 (||X||^2 - 2<X,UV^T> + <U^TU,V^TV>)
    double reg_err = 0.0, err = rec;
    if (st.mode == BMF_MODE_PENALTY) {
        reg_err = reg_used * (0.5 * st.comm[1] + 0.5 * st.scal[0]);
        err = rec + reg_err;
    }
    const double tp = st.comm[2], fp = st.comm[3];
    const double fn = st.sum_x - tp;
    const double tn = st.cells - tp - fp - fn;
    double* row = st.log + (int64_t)iter * BMF_LOG_COLS;
    row[BMF_LOG_ITER] = (double)iter;
    row[BMF_LOG_ERROR] = err;
    row[BMF_LOG_REC] = rec;
    row[BMF_LOG_REG] = reg_used;
    row[BMF_LOG_REGERR] = reg_err;
    row[BMF_LOG_RMSE] = sqrt(fmax(2.0 * rec, 0.0) / st.cells);
    row[BMF_LOG_MAE] = st.with_mae ? st.comm[4] / st.cells : __builtin_nan("");
    row[BMF_LOG_TP] = tp;
    row[BMF_LOG_FP] = fp;
    row[BMF_LOG_FN] = fn;
    row[BMF_LOG_TN] = tn;
    row[BMF_LOG_VALID] = 1.0;
    // early stop: BinaryMFPenalty watches reg_error (BinaryMFPenalty.py:89,112), WNMF the error (WNMF.py:72,89)
    const double watched = st.mode == BMF_MODE_PENALTY ? reg_err : err;
    int stop_now = 0;
    if (iter >= 1) {
        const double diff = fabs(st.scal[1] - watched);
        if (watched <= st.tol) stop_now = 1;
        if (iter > max_iter) stop_now = 1;
        if (diff < st.min_diff) stop_now = 1;
    }
    st.scal[1] = watched;
    row[BMF_LOG_STOP] = (double)stop_now;
    if (stop_now) *st.stop = iter;
}

__global__ __launch_bounds__(1024) void finalize_kernel(bmf_penalty_state st, int iter, double reg_used, int max_iter, int do_gather) {
    __shared__ double sh[1024];
    finalize_body(st, iter, reg_used, max_iter, do_gather, sh);
}

// The single-GPU loop's last two launches of an iteration as one: block 0 writes the log row, blocks [1, red_blocks] sum the X^T U slabs into Nred (the
// arithmetic of reduce_slabs_wide_kernel: every element over the slabs in slab order).  The log
// row needs nothing of X^T U (the numerator of the NEXT V update): the Gram matrices, the cover counts and the partial sums were
// complete before that GEMM started.  A stop flag raised here races with the reduction blocks beside it, harmlessly: Nred is only
// read by an update that the flag cancels.  (~8 us of stream time per iteration: a launch of its own costs that much.)
__global__ __launch_bounds__(1024) void reduce_finalize_kernel(bmf_penalty_state st, int iter, double reg_used, int max_iter, int red_blocks, int count) {
    __shared__ double sh[1024];
    if (blockIdx.x == 0) {   // dispatched first: the log row is the longer dependent chain of the two
        finalize_body(st, iter, reg_used, max_iter, 1, sh);
        return;
    }
    if (*st.stop != 0) return;
    const int rb = (int)blockIdx.x - 1;
    const int64_t stride = st.n_pad * st.kp, n4 = stride / 4;
    for (int64_t i = (int64_t)rb * 1024 + threadIdx.x; i < n4; i += (int64_t)red_blocks * 1024) {
        const float* p = st.Nslab + 4 * i;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int b = 0;
        for (; b + 4 <= count; b += 4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + stride);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + 2 * stride);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + 3 * stride);
            p += 4 * stride;
            a0 = (((a0 + (double)v0[0]) + (double)v1[0]) + (double)v2[0]) + (double)v3[0];
            a1 = (((a1 + (double)v0[1]) + (double)v1[1]) + (double)v2[1]) + (double)v3[1];
            a2 = (((a2 + (double)v0[2]) + (double)v1[2]) + (double)v2[2]) + (double)v3[2];
            a3 = (((a3 + (double)v0[3]) + (double)v1[3]) + (double)v2[3]) + (double)v3[3];
        }
        for (; b < count; ++b) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p);
            p += stride;
            a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
        }
        *reinterpret_cast<f32x4*>(st.Nred + 4 * i) = f32x4{(float)a0, (float)a1, (float)a2, (float)a3};
    }
}

}  // namespace

static int check_state(const bmf_penalty_state* st, const char* who) {
    BMF_REQUIRE(st, "%s: null state", who);
    BMF_REQUIRE(st->struct_bytes == (int32_t)sizeof(bmf_penalty_state), "%s: struct_bytes=%d, library expects %d", who,
                st->struct_bytes, (int)sizeof(bmf_penalty_state));
    BMF_REQUIRE(st->m >= 1 && st->n >= 1 && st->k >= 1, "%s: m, n, k must be positive", who);
    BMF_REQUIRE((st->kp == 32 || st->kp == 64) && st->k <= st->kp, "%s: kp must be 32 or 64 and >= k (k <= 64 supported)", who);
    BMF_REQUIRE(st->terms >= 1 && st->terms <= 3, "%s: terms must be 1..3", who);
    BMF_REQUIRE(st->mode == BMF_MODE_PENALTY || st->mode == BMF_MODE_WNMF, "%s: mode must be PENALTY or WNMF", who);
    BMF_REQUIRE(st->m_pad % BMF_ROW_PAD == 0 && st->n_pad % BMF_ROW_PAD == 0 && st->m_pad >= st->m && st->n_pad >= st->n,
                "%s: m_pad/n_pad must be multiples of %d covering m/n", who, BMF_ROW_PAD);
    BMF_REQUIRE(st->ldx == st->n_pad / 32 && st->ldxt == st->m_pad / 32, "%s: ldx must be n_pad/32 and ldxt m_pad/32", who);
    BMF_REQUIRE(st->Xbits && st->XTbits && st->U64 && st->V64 && st->U && st->V && st->Upanel && st->Vpanel && st->Mslab && st->Nslab && st->Nred &&
                    st->gram_slabs && st->GU && st->GV && st->comm && st->GV64 && st->partU && st->partV && st->scal &&
                    st->ubits && st->ucolbits && st->vbits && st->vcolbits && st->counts && st->log && st->stop,
                "%s: null device pointer in state", who);
    BMF_REQUIRE(st->splits_xv >= 1 && st->splits_xtu >= 1, "%s: splits must be >= 1", who);
    BMF_REQUIRE(st->nred_blocks == 0 || st->nred_blocks == 1 || (st->nred_blocks == 2 && st->kp == 64 && st->panel_kind == BMF_PANEL_I8),
                "%s: nred_blocks must be 0 / 1, or 2 with kp == 64 and BMF_PANEL_I8", who);
    BMF_REQUIRE(st->gram_blocks >= 1 && st->gram_blocks <= 1024, "%s: gram_blocks must be 1..1024", who);
    BMF_REQUIRE(st->lduc >= st->m_pad / 32 && st->ldvc >= st->n_pad / 32, "%s: lduc/ldvc too small", who);
    BMF_REQUIRE(st->log_rows >= 1, "%s: log_rows must be >= 1", who);
    BMF_REQUIRE(st->panel_kind == BMF_PANEL_BF16 ||
                    (st->panel_kind == BMF_PANEL_F16 && st->terms == 2 && st->scaleU && st->scaleV && st->panel_ws) ||
                    (st->panel_kind == BMF_PANEL_I8 && st->terms >= 2 && st->scaleU && st->scaleV && st->panel_ws),
                "%s: panel_kind must be BMF_PANEL_BF16, BMF_PANEL_F16 (terms == 2) or BMF_PANEL_I8 (terms 2 or 3), the last two with "
                "scaleU, scaleV and panel_ws", who);
    return BMF_OK;
}

#define BMF_TRY(expr)            \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != BMF_OK) return rc_; \
    } while (0)

// out[b][row][c] = sum_s slabs[s][row][32 b + c]: the X^T U slabs of column block b, summed in slab order, into the
// block-major exchange buffer
namespace {
__global__ __launch_bounds__(256) void reduce_slabs_block_kernel(const float* __restrict__ slabs, int64_t slab_stride, int count, int64_t rows_pad,
                                                                  int kp, int block, float* __restrict__ out, const int32_t* __restrict__ stop) {
    const bool stopped = stop && *stop != 0;   // (then the block is zeroed: see gather_kernel)
    const int64_t total = rows_pad * 8;  // float4 pieces of the block
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i >> 3;
        const int c4 = (int)(i & 7) * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < (stopped ? 0 : count); ++sp) acc += *reinterpret_cast<const f32x4*>(slabs + (int64_t)sp * slab_stride + row * kp + 32 * block + c4);
        *reinterpret_cast<f32x4*>(out + ((int64_t)block * rows_pad + row) * 32 + c4) = acc;
    }
}
}  // namespace

// slab slots the X^T U launch of this state fills (<= splits_xtu, the size of the slab array)
static int xtu_slots_used(const bmf_penalty_state* st) {
    int n = st->splits_xtu;
    if (st->panel_kind == BMF_PANEL_I8) {
        const int need = bmf_xf_bits_i8_slots(st->n_pad, st->m_pad / 32, st->nred_blocks == 2 ? 32 : st->kp);
        if (need >= 1 && need < n) n = need;
    }
    return n;
}

// one sweep in two phases.  HEAD: V epilogue, V^T V, X V, U epilogue, then the scalar part (U^T U, cover count, MAE, gather ->
// comm, the fp64 exchange buffer).  XTU: X^T U of the new U (-> Nred, the fp32 exchange buffer), whole or one 32-column block.
// When sharded, the caller starts the all-reduce of a block as soon as it is enqueued, so that it overlaps the next block's
// GEMM.  mode = PREPARE for iteration 0.
// SCALARS: what of the new (U, V) goes into the fp64 exchange block besides U^T U -- cover count, MAE sums, gather -> comm.  In the
// single-GPU loop it follows the head; in the sharded loop it runs AFTER the X^T U GEMM, under the all-reduce of the numerator.
enum { SWEEP_HEAD = 1, SWEEP_XTU = 2, SWEEP_SCALARS = 4, SWEEP_ALL = 7 };
static int sweep(const bmf_penalty_state* st, int mode, double reg, hipStream_t s, int phase = SWEEP_ALL, int block = -1, bool gather_in_finalize = false,
                 bool reduce_in_finalize = false) {
    const int kp = st->kp, kk = kp * kp;
    const int32_t* stop = st->stop;
    const bool f16 = st->panel_kind == BMF_PANEL_F16, i8 = st->panel_kind == BMF_PANEL_I8;
    const bool blocked = st->nred_blocks == 2;
    // (A/B switch for measurements: BMF_I8_FUSED_PLANES=0 builds the int8 planes with the stand-alone kernel every iteration, as
    // before round 3)
    static const bool fused_planes = [] { const char* e = getenv("BMF_I8_FUSED_PLANES"); return !(e && e[0] == '0'); }();
    // fp16 / int8 panels need the column maxima of the whole updated factor, so they are built after the epilogue (which
    // then builds no panel of its own: terms = 0)
    const int epi_terms = (f16 || i8) ? 0 : st->terms;

    if (phase & SWEEP_HEAD) {
        bmf_epilogue_args ev = {};
        ev.F64 = st->V64; ev.F = st->V; ev.rows_pad = st->n_pad; ev.rows = st->n; ev.k = st->k; ev.kp = kp;
        ev.num = mode == BMF_MODE_PREPARE ? nullptr : st->Nred; ev.slab_stride = st->n_pad * kp; ev.splits = 1;
        ev.num_block_stride = blocked ? st->n_pad * 32 : 0;
        ev.G = st->GU; ev.reg = reg; ev.mode = mode; ev.thr = st->thr_v; ev.terms = epi_terms;
        ev.panel = st->Vpanel; ev.ldp = st->n_pad; ev.rowbits = st->vbits; ev.colbits = st->vcolbits; ev.ldcb = st->ldvc;
        ev.partials = st->partV; ev.stop = stop; ev.blockmax = (f16 || i8) ? st->panel_ws : nullptr;
        // int8 planes: emitted by the epilogue itself with the scale predicted from the previous iteration's column maxima; the
        // column-scale blocks of the Gram launch check the prediction and the builder below rebuilds only when it was off
        if (i8 && fused_planes) { ev.planes = (int8_t*)st->Vpanel; ev.plane_scale = st->scaleV; ev.limbs = st->terms; }
        BMF_TRY(bmf_mu_epilogue(&ev, s));
        if (f16) BMF_TRY(bmf_panel_f16_launch(st->V, st->n_pad, kp, kp, st->Vpanel, st->n_pad, st->panel_ws, st->scaleV, true, stop, s));
        // int8 planes: the column scales are derived by extra blocks of the Gram launch (one launch less in the chain)
        BMF_TRY(bmf_gram_partial_launch(st->V, st->n_pad, kp, kp, st->gram_slabs, st->gram_blocks, i8 ? st->panel_ws : nullptr, st->terms, st->scaleV, stop, s, fused_planes ? 1 : 0));
        // (fused: the conditional rebuild of the planes and the reduction of the Gram slabs are ONE launch)
        if (i8 && fused_planes)
            BMF_TRY(bmf_panel_i8_launch(st->V64, st->V, st->n_pad, kp, kp, st->terms, (int8_t*)st->Vpanel, st->n_pad, st->panel_ws, st->scaleV + 2 * kp, true, stop, s, true,
                                        st->scaleV + 3 * kp, st->scaleV + kp, st->gram_slabs, st->gram_blocks, kk, st->GV, st->GV64));
        else {
            if (i8)
                BMF_TRY(bmf_panel_i8_launch(st->V64, st->V, st->n_pad, kp, kp, st->terms, (int8_t*)st->Vpanel, st->n_pad, st->panel_ws, st->scaleV, true, stop, s, true, nullptr, nullptr,
                                            nullptr, 0, 0, nullptr, nullptr));
            BMF_TRY(bmf_reduce_slabs(st->gram_slabs, kk, st->gram_blocks, kk, st->GV, st->GV64, s));
        }

        bmf_timer_begin(s);
        if (i8)
            BMF_TRY(bmf_xf_bits_i8_launch(st->Xtiled ? st->Xtiled : st->Xbits, st->m_pad, st->ldx, st->n_pad / 32, (const int8_t*)st->Vpanel,
                                          st->n_pad, st->terms, st->scaleV + kp, kp, 0, kp, st->Mslab, st->m_pad * kp, st->splits_xv,
                                          st->Xtiled != nullptr, stop, s));
        else
            BMF_TRY(bmf_xf_bits_launch(st->Xbits, st->m_pad, st->ldx, st->n_pad / 32, st->Vpanel, st->n_pad, st->terms, kp, st->Mslab,
                                       st->m_pad * kp, st->splits_xv, st->panel_kind, f16 ? st->scaleV + kp : nullptr, stop, s));
        bmf_timer_end(s);

        bmf_epilogue_args eu = {};
        eu.F64 = st->U64; eu.F = st->U; eu.rows_pad = st->m_pad; eu.rows = st->m; eu.k = st->k; eu.kp = kp;
        // A short shard is cut into many stream-K slices per row tile: the epilogue (128-row blocks, few of them) would read all
        // those partial slabs with a handful of workgroups.  Sum them first with the whole chip, in place into slab 0 (row-sharded
        // runs: 25 -> 18 us at m = 12 500; at full size there are two slabs and this does not apply).
        const bool presum = st->splits_xv >= 6;
        if (presum)
            BMF_TRY(bmf_reduce_slabs(st->Mslab, st->m_pad * kp, st->splits_xv, st->m_pad * kp, st->Mslab, nullptr, s));
        eu.num = st->Mslab; eu.slab_stride = st->m_pad * kp; eu.splits = presum ? 1 : st->splits_xv;
        eu.G = st->GV; eu.reg = reg; eu.mode = mode; eu.thr = st->thr_u; eu.terms = epi_terms;
        eu.panel = st->Upanel; eu.ldp = st->m_pad; eu.rowbits = st->ubits; eu.colbits = st->ucolbits; eu.ldcb = st->lduc;
        eu.partials = st->partU; eu.stop = stop; eu.blockmax = (f16 || i8) ? st->panel_ws : nullptr;
        if (i8 && fused_planes) { eu.planes = (int8_t*)st->Upanel; eu.plane_scale = st->scaleU; eu.limbs = st->terms; }
        BMF_TRY(bmf_mu_epilogue(&eu, s));
        if (f16) BMF_TRY(bmf_panel_f16_launch(st->U, st->m_pad, kp, kp, st->Upanel, st->m_pad, st->panel_ws, st->scaleU, true, stop, s));
        BMF_TRY(bmf_gram_partial_launch(st->U, st->m_pad, kp, kp, st->gram_slabs, st->gram_blocks, i8 ? st->panel_ws : nullptr, st->terms, st->scaleU, stop, s, fused_planes ? 1 : 0));
        // the scalar part (everything of the new (U, V) that goes into the fp64 exchange block) starts with U^T U: its slab reduction
        // rides in the launch of the conditional plane rebuild when there is one
        if (i8 && fused_planes)
            BMF_TRY(bmf_panel_i8_launch(st->U64, st->U, st->m_pad, kp, kp, st->terms, (int8_t*)st->Upanel, st->m_pad, st->panel_ws, st->scaleU + 2 * kp, true, stop, s, true,
                                        st->scaleU + 3 * kp, st->scaleU + kp, st->gram_slabs, st->gram_blocks, kk, nullptr, st->comm + 8));
        else {
            if (i8)
                BMF_TRY(bmf_panel_i8_launch(st->U64, st->U, st->m_pad, kp, kp, st->terms, (int8_t*)st->Upanel, st->m_pad, st->panel_ws, st->scaleU, true, stop, s, true, nullptr, nullptr,
                                            nullptr, 0, 0, nullptr, nullptr));
            BMF_TRY(bmf_reduce_slabs(st->gram_slabs, kk, st->gram_blocks, kk, nullptr, st->comm + 8, s));
        }
    }
    if (phase & SWEEP_SCALARS) {
        if (!st->updates_only)
            BMF_TRY(bmf_cover_launch(st->Xbits, st->m_pad, st->ldx, st->n_pad / 32, st->ubits, st->vcolbits, st->ldvc, kp, st->counts,
                                     stop, s));
        if (st->with_mae && !st->updates_only) {
            const bool one_product = st->m_pad * st->n_pad >= (1 << 24);   // then the operand conversion zeroes comm[4], comm[5] itself
            if (!(st->mae_ws && one_product)) BMF_LAUNCH(zero_mae_kernel, dim3(1), dim3(64), 0, s, st->comm, stop);
            if (st->mae_ws) {
                // the tiled copy of X^T (when the int8 GEMM has one) streams better: one 4-KiB piece per stage instead of 64 row pieces
                const bool xt = st->XTtiled && st->m_pad * st->n_pad >= (1 << 24) && st->n_pad % 256 == 0 && st->ldxt % 16 == 0 && st->ldxt * 32 == st->m_pad;
                BMF_TRY(bmf_mae_launch(xt ? st->XTtiled : st->XTbits, st->ldxt, st->m_pad, st->n_pad, st->U, st->V, kp, st->mae_ws, st->comm + 4, stop, s,
                                       -1, xt ? 1 : 0, one_product ? 1 : 0));
            }
            else
                BMF_TRY(bmf_residual_launch(st->Xbits, st->m_pad, st->ldx, st->m, st->n, st->U, st->V, nullptr, nullptr, kp,
                                            st->comm + 4, stop, s));
        }
        if (!gather_in_finalize)
            BMF_LAUNCH(gather_kernel, dim3(1), dim3(256), 0, s, st->partU, (int)(st->m_pad / 128), st->partV,
                               (int)(st->n_pad / 128), st->counts, st->comm, st->scal, stop);
        BMF_LAUNCH_CHECK();
    }
    if (!(phase & SWEEP_XTU)) return BMF_OK;

    // X^T U of the new U: the numerator of the NEXT V update.  The slab array is sized for the launch form that needs most slots (32-column
    // blocks: 10 at the headline shape), the whole-factor launch needs fewer (6): fill and sum only those -- the rest was 21 MB of zeros
    // written and 21 MB read per iteration.
    const int xtu_slots = xtu_slots_used(st);
    const int b0 = block < 0 ? 0 : block, b1 = block < 0 ? (blocked ? 2 : 1) : block + 1;
    for (int b = b0; b < b1; ++b) {
        bmf_timer_begin(s);
        if (i8)
            BMF_TRY(bmf_xf_bits_i8_launch(st->XTtiled ? st->XTtiled : st->XTbits, st->n_pad, st->ldxt, st->m_pad / 32, (const int8_t*)st->Upanel,
                                          st->m_pad, st->terms, st->scaleU + kp, kp, blocked ? 32 * b : 0, blocked ? 32 : kp, st->Nslab,
                                          st->n_pad * kp, xtu_slots, st->XTtiled != nullptr, stop, s));
        else
            BMF_TRY(bmf_xf_bits_launch(st->XTbits, st->n_pad, st->ldxt, st->m_pad / 32, st->Upanel, st->m_pad, st->terms, kp, st->Nslab,
                                       st->n_pad * kp, st->splits_xtu, st->panel_kind, f16 ? st->scaleU + kp : nullptr, stop, s));
        bmf_timer_end(s);
        if (blocked) {
            const int64_t pieces = st->n_pad * 8;
            BMF_LAUNCH(reduce_slabs_block_kernel, dim3((unsigned)((pieces + 255) / 256 < 2048 ? (pieces + 255) / 256 : 2048)), dim3(256), 0, s,
                       st->Nslab, st->n_pad * kp, xtu_slots, st->n_pad, kp, b, st->Nred, stop);
            BMF_LAUNCH_CHECK();
        } else if (!reduce_in_finalize) {
            BMF_TRY(bmf_reduce_slabs(st->Nslab, st->n_pad * kp, xtu_slots, st->n_pad * kp, st->Nred, nullptr, s));
        }
    }
    return BMF_OK;
}

extern "C" int bmf_penalty_prepare(const bmf_penalty_state* st, void* stream) {
    BMF_TRY(check_state(st, "bmf_penalty_prepare"));
    return sweep(st, BMF_MODE_PREPARE, 0.0, (hipStream_t)stream);
}

extern "C" int bmf_penalty_update(const bmf_penalty_state* st, double reg, void* stream) {
    BMF_TRY(check_state(st, "bmf_penalty_update"));
    return sweep(st, st->mode, reg, (hipStream_t)stream);
}

extern "C" int bmf_penalty_update_head(const bmf_penalty_state* st, double reg, void* stream) {
    BMF_TRY(check_state(st, "bmf_penalty_update_head"));
    return sweep(st, st->mode, reg, (hipStream_t)stream, SWEEP_HEAD | SWEEP_SCALARS);
}

extern "C" int bmf_penalty_update_xtu(const bmf_penalty_state* st, int32_t block, void* stream) {
    BMF_TRY(check_state(st, "bmf_penalty_update_xtu"));
    BMF_REQUIRE(block >= -1 && block < (st->nred_blocks == 2 ? 2 : 1), "bmf_penalty_update_xtu: block=%d out of range", block);
    return sweep(st, st->mode, 0.0, (hipStream_t)stream, SWEEP_XTU, block);
}

extern "C" int bmf_penalty_finalize(const bmf_penalty_state* st, int32_t iter, double reg_used, int32_t max_iter,
                                    void* stream) {
    BMF_TRY(check_state(st, "bmf_penalty_finalize"));
    BMF_REQUIRE(iter >= 0 && iter < st->log_rows, "bmf_penalty_finalize: iter=%d outside the %d-row log", iter, st->log_rows);
    BMF_LAUNCH(finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, *st, (int)iter, reg_used, (int)max_iter, 0);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_penalty_run(const bmf_penalty_state* st, int32_t iter0, int32_t iter1, const double* regs_host,
                               int32_t max_iter, void* stream) {
    BMF_TRY(check_state(st, "bmf_penalty_run"));
    BMF_REQUIRE(regs_host, "bmf_penalty_run: null regs_host");
    BMF_REQUIRE(iter0 >= 1 && iter1 >= iter0 && iter1 <= st->log_rows, "bmf_penalty_run: bad iteration range [%d,%d) for %d log rows",
                iter0, iter1, st->log_rows);
    for (int it = iter0; it < iter1; ++it) {
        const double reg = regs_host[it - iter0];
        // (the slab reduction and the log row as ONE launch: 12.6 us against 10.7 + 10.3)
        const bool fuse = st->nred_blocks != 2 && st->n_pad * st->kp >= 65536 && bmf_aligned16(st->Nslab) && bmf_aligned16(st->Nred);
        BMF_TRY(sweep(st, st->mode, reg, (hipStream_t)stream, SWEEP_ALL, -1, true, fuse));
        if (fuse) {
            const int64_t n4 = st->n_pad * st->kp / 4;
            const int red_blocks = (int)((n4 + 1023) / 1024 < 1024 ? (n4 + 1023) / 1024 : 1024);
            BMF_LAUNCH(reduce_finalize_kernel, dim3((unsigned)red_blocks + 1), dim3(1024), 0, (hipStream_t)stream, *st, it, reg, (int)max_iter, red_blocks,
                       xtu_slots_used(st));
        } else {
            BMF_LAUNCH(finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, *st, it, reg, (int)max_iter, 1);
        }
        BMF_LAUNCH_CHECK();
    }
    return BMF_OK;
}

// ---------------------------------------------------------------------------------------------------
// the row-sharded loop, collectives included, enqueued from C (reference loop body: BinaryMFPenalty.py:81-115)
// ---------------------------------------------------------------------------------------------------
namespace {

// compute stream s -> side stream: "what s has enqueued so far is complete" (event slot e)
int fence_to_comm(bmf_comm* c, int e, hipStream_t s) {
    BMF_HIP_CHECK(hipEventRecord(c->ev[e], s));
    BMF_HIP_CHECK(hipStreamWaitEvent(c->cs, c->ev[e], 0));
    return BMF_OK;
}

// With more than one rank the all-reduce of the numerator (n_pad x kp fp32: 5.2 MB at the headline shape) is the long pole of the
// exchange, and the scalar part of the step -- cover count, MAE sums, gather: everything of the new (U, V) that goes into the fp64
// block besides U^T U -- depends on nothing the X^T U GEMM produces.  It can be enqueued BEHIND that GEMM, on the compute stream,
// while the numerator's all-reduce runs on the side stream: it then hides under the exchange instead of standing in front of it,
// and the small fp64 all-reduce follows it.  The price is three stream crossings (event record + wait) of ~12 us each, measured
// with one rank, where the collectives are free: 0.206 ms per step against 0.171 in stream order at 12 500 rows
// (profiles/r04_shard_sizes.txt).  So the scalar part goes under the exchange only where it is longer than that: with the MAE pass
// in the step (0.27 ms at 100 000 rows, the default of the model classes) or from 65 536 rows per rank on (cover count >= 35 us);
// below, and always with one rank, everything stays in stream order.  BMF_EXCHANGE_OVERLAP=0|1 overrides.
// The two forms issue DIFFERENT collectives (order, grouping), so every rank must take the same one: shards differ by up to 32 rows
// and pad to 512, i.e. the local m_pad can straddle a threshold (world 2, m in (15360, 15392]: 8192 and 7680).  The caller that
// knows all shards says so in st->exchange_overlap (engine.MUEngine: from the LARGEST shard, one all-reduce(MAX) at construction);
// the rule on the local m_pad below is for states that leave it 0 -- one rank, or shards known to be equal.
static bool overlap_exchange(const bmf_penalty_state* st, const bmf_comm* c) {
    static const int env = [] { const char* e = getenv("BMF_EXCHANGE_OVERLAP"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    if (env >= 0) return env == 1;
    if (c->world <= 1 || st->updates_only) return false;
    if (st->exchange_overlap != 0) return st->exchange_overlap == 2;
    return bmf_exchange_overlap_rule(st->with_mae, st->m_pad) != 0;
}

// X^T U of the new U, the scalar part and the exchange of one iteration.  On entry the head (through the U side's digit planes and
// U^T U) has been enqueued on s; on return s has been told to wait for the collectives.
int exchange_phase(const bmf_penalty_state* st, bmf_comm* c, hipStream_t s) {
    const int kp = st->kp;
    const int64_t n32 = st->n_pad * kp, n64 = 8 + (int64_t)kp * kp;
    const bool timed = c->t_used < c->t_cap;
    const bool overlap = overlap_exchange(st, c);
    if (!overlap) BMF_TRY(sweep(st, st->mode, 0.0, s, SWEEP_SCALARS));   // in front of the GEMM, as the single-GPU loop has it
    if (timed) BMF_HIP_CHECK(hipEventRecord(c->tev[3 * c->t_used], s));
    if (st->nred_blocks == 2) {
        // block 0, then its all-reduce under the GEMM of block 1, whose all-reduce runs under the scalar part
        BMF_TRY(sweep(st, st->mode, 0.0, s, SWEEP_XTU, 0));
        BMF_TRY(fence_to_comm(c, 0, s));
        if (overlap) {
            BMF_TRY(bmf_comm_allreduce_on(c, st->Nred, n32 / 2, BMF_DTYPE_F32, c->cs));
        } else {
            BMF_TRY(bmf_comm_group_begin(c));
            int rc = bmf_comm_allreduce_on(c, st->Nred, n32 / 2, BMF_DTYPE_F32, c->cs);
            if (rc == BMF_OK) rc = bmf_comm_allreduce_on(c, st->comm, n64, BMF_DTYPE_F64, c->cs);
            const int rce = bmf_comm_group_end(c);
            if (rc != BMF_OK || rce != BMF_OK) return rc != BMF_OK ? rc : rce;
        }
        BMF_TRY(sweep(st, st->mode, 0.0, s, SWEEP_XTU, 1));
        BMF_TRY(fence_to_comm(c, 1, s));
        BMF_TRY(bmf_comm_allreduce_on(c, st->Nred + n32 / 2, n32 / 2, BMF_DTYPE_F32, c->cs));
        if (overlap) {
            BMF_TRY(sweep(st, st->mode, 0.0, s, SWEEP_SCALARS));
            BMF_TRY(fence_to_comm(c, 3, s));
            BMF_TRY(bmf_comm_allreduce_on(c, st->comm, n64, BMF_DTYPE_F64, c->cs));
        }
    } else if (overlap) {
        BMF_TRY(sweep(st, st->mode, 0.0, s, SWEEP_XTU, -1));
        BMF_TRY(fence_to_comm(c, 0, s));
        BMF_TRY(bmf_comm_allreduce_on(c, st->Nred, n32, BMF_DTYPE_F32, c->cs));
        BMF_TRY(sweep(st, st->mode, 0.0, s, SWEEP_SCALARS));
        BMF_TRY(fence_to_comm(c, 1, s));
        BMF_TRY(bmf_comm_allreduce_on(c, st->comm, n64, BMF_DTYPE_F64, c->cs));
    } else {
        // One rank, one launch of X^T U: nothing to hide anything under, so the whole exchange -- numerator and scalars, ONE grouped
        // RCCL launch -- goes on the compute stream itself, in order.  No side stream, no events (measured with one rank, where the
        // collective itself is free: 0.222 vs 0.183 ms per step at 12 500 rows with the fences).
        BMF_TRY(sweep(st, st->mode, 0.0, s, SWEEP_XTU, -1));
        if (timed) BMF_HIP_CHECK(hipEventRecord(c->tev[3 * c->t_used + 1], s));
        BMF_TRY(bmf_allreduce(c, st->Nred, n32, st->comm, n64, s));
        if (timed) {
            BMF_HIP_CHECK(hipEventRecord(c->tev[3 * c->t_used + 2], s));
            ++c->t_used;
        }
        return BMF_OK;
    }
    BMF_HIP_CHECK(hipEventRecord(c->ev[2], c->cs));
    if (timed) BMF_HIP_CHECK(hipEventRecord(c->tev[3 * c->t_used + 1], s));
    BMF_HIP_CHECK(hipStreamWaitEvent(s, c->ev[2], 0));
    if (timed) {
        BMF_HIP_CHECK(hipEventRecord(c->tev[3 * c->t_used + 2], s));
        ++c->t_used;
    }
    return BMF_OK;
}

int check_comm(const bmf_penalty_state* st, const bmf_comm* c, const char* who) {
    BMF_REQUIRE(c && c->cs && (c->kind == BMF_COMM_RCCL || c->kind == BMF_COMM_HOST), "%s: null or uninitialised communicator", who);
    BMF_REQUIRE(st->nred_blocks <= 2, "%s: nred_blocks=%d", who, st->nred_blocks);
    return BMF_OK;
}

}  // namespace

// the shard-size rule behind that decision, for callers that evaluate it on a rank-invariant row count (the largest shard's m_pad)
extern "C" int bmf_exchange_overlap_rule(int with_mae, int64_t m_pad) { return ((with_mae && m_pad >= 8192) || m_pad >= 65536) ? 1 : 0; }

// 1 when bmf_penalty_run_sharded places the scalar part of a step under the numerator's all-reduce for this state on this communicator
extern "C" int bmf_exchange_overlaps(const bmf_penalty_state* st, const bmf_comm* comm) {
    BMF_REQUIRE(st && comm, "bmf_exchange_overlaps: null pointer");
    return overlap_exchange(st, comm) ? 1 : 0;
}

extern "C" int bmf_penalty_prepare_sharded(const bmf_penalty_state* st, bmf_comm* comm, double reg0, int32_t max_iter, void* stream) {
    BMF_TRY(check_state(st, "bmf_penalty_prepare_sharded"));
    BMF_TRY(check_comm(st, comm, "bmf_penalty_prepare_sharded"));
    hipStream_t s = (hipStream_t)stream;
    // iteration-0 bookkeeping: the same products as an update (BinaryMFPenalty.py:68-75), then the blocking form of the exchange
    BMF_TRY(sweep(st, BMF_MODE_PREPARE, 0.0, s));
    const int64_t n32 = st->n_pad * st->kp, n64 = 8 + (int64_t)st->kp * st->kp;
    if (st->nred_blocks == 2) {
        // block by block, as every later iteration does it: an all-reduce over more than two ranks adds in an order that depends on
        // how the buffer is cut into chunks, and the two blocks as ONE buffer would be cut differently (the C loop and the
        // host-driven protocol must agree bit for bit: tests/test_sharded_gpu.py)
        BMF_TRY(bmf_comm_group_begin(comm));
        int rc = bmf_comm_allreduce_on(comm, st->Nred, n32 / 2, BMF_DTYPE_F32, s);
        if (rc == BMF_OK) rc = bmf_comm_allreduce_on(comm, st->Nred + n32 / 2, n32 / 2, BMF_DTYPE_F32, s);
        if (rc == BMF_OK) rc = bmf_comm_allreduce_on(comm, st->comm, n64, BMF_DTYPE_F64, s);
        const int rce = bmf_comm_group_end(comm);
        if (rc != BMF_OK || rce != BMF_OK) return rc != BMF_OK ? rc : rce;
    } else {
        BMF_TRY(bmf_allreduce(comm, st->Nred, n32, st->comm, n64, s));
    }
    BMF_LAUNCH(finalize_kernel, dim3(1), dim3(1024), 0, s, *st, 0, reg0, (int)max_iter, 0);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_penalty_run_sharded(const bmf_penalty_state* st, bmf_comm* comm, int32_t iter0, int32_t iter1, const double* regs_host,
                                       int32_t max_iter, void* stream) {
    BMF_TRY(check_state(st, "bmf_penalty_run_sharded"));
    BMF_TRY(check_comm(st, comm, "bmf_penalty_run_sharded"));
    BMF_REQUIRE(regs_host, "bmf_penalty_run_sharded: null regs_host");
    BMF_REQUIRE(iter0 >= 1 && iter1 >= iter0 && iter1 <= st->log_rows, "bmf_penalty_run_sharded: bad iteration range [%d,%d) for %d log rows",
                iter0, iter1, st->log_rows);
    hipStream_t s = (hipStream_t)stream;
    for (int it = iter0; it < iter1; ++it) {
        const double reg = regs_host[it - iter0];
        BMF_TRY(sweep(st, st->mode, reg, s, SWEEP_HEAD));   // (the scalar part is placed by exchange_phase)
        BMF_TRY(exchange_phase(st, comm, s));
        BMF_LAUNCH(finalize_kernel, dim3(1), dim3(1024), 0, s, *st, it, reg, (int)max_iter, 0);
        BMF_LAUNCH_CHECK();
    }
    return BMF_OK;
}
