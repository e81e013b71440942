// loss_kernels.hip -- training-side hot path.
//
//   se_cosine_loss_fwd : l2norm head + class-embedding gather + inv_correlation + batch mean
//                        (reference: utils.py:125-127, learn_image_embeddings.py:48-50,127-128,
//                         utils.py:44-46)
//   se_cosine_loss_bwd : closed-form backward of the above
//   se_nn_accuracy     : nearest-class-embedding accuracy, y_pred @ emb^T on
//                        v_mfma_f32_32x32x2_f32 (reference: utils.py:57-100)
//
// Layout: one 64-lane wavefront owns one feature row; rows are streamed with coalesced
// 16-byte loads (float4 / 8 x bf16) when the row pitch allows it, the two row reductions
// (sum x^2 and x . emb[y]) are done with DPP/shuffle wave reductions -- no LDS, no atomics.
// The kernels are HBM-bound: algorithmic bytes per row = D*s_x (x) + D*4 (emb row, L2-resident)
// + D*4 (xhat) + 16.
#include "se_common.h"

namespace se {

constexpr int LOSS_ROWS_PER_BLOCK = 4;  // 4 waves = 256 threads
constexpr float L2NORM_EPS = 1e-12f;    // tf.nn.l2_normalize default epsilon (TF 1.x)

template <bool BF16>
__device__ __forceinline__ float load_x(const void *row, int64_t d)
{
    if constexpr (BF16) return bf16_to_f32(((const uint16_t *)row)[d]);
    else return ((const float *)row)[d];
}

// ss = sum x^2, dt = sum x * t  over one row, all 64 lanes get the totals.
template <bool BF16>
__device__ __forceinline__ void row_reduce(const void *xrow, const float *trow, int64_t D, bool vec_ok,
                                           float &ss, float &dt)
{
    const int lane = lane_id();
    float s = 0.f, t = 0.f;
    if (vec_ok) {
        if constexpr (BF16) {
            const uint4 *xv = (const uint4 *)xrow;  // 8 bf16 per 16 B
            const float4 *tv = (const float4 *)trow;
            for (int64_t i = lane; i < D / 8; i += WAVE) {
                uint4 p = xv[i];
                float4 t0 = tv[2 * i], t1 = tv[2 * i + 1];
                uint32_t w[4] = {p.x, p.y, p.z, p.w};
                float xs[8];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    xs[2 * j] = __uint_as_float(w[j] << 16);
                    xs[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
                }
                float ts[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    s = fmaf(xs[j], xs[j], s);
                    t = fmaf(xs[j], ts[j], t);
                }
            }
        } else {
            const float4 *xv = (const float4 *)xrow;
            const float4 *tv = (const float4 *)trow;
            for (int64_t i = lane; i < D / 4; i += WAVE) {
                float4 p = xv[i], q = tv[i];
                s = fmaf(p.x, p.x, s); s = fmaf(p.y, p.y, s); s = fmaf(p.z, p.z, s); s = fmaf(p.w, p.w, s);
                t = fmaf(p.x, q.x, t); t = fmaf(p.y, q.y, t); t = fmaf(p.z, q.z, t); t = fmaf(p.w, q.w, t);
            }
        }
    } else {
        for (int64_t i = lane; i < D; i += WAVE) {
            float p = load_x<BF16>(xrow, i);
            s = fmaf(p, p, s);
            t = fmaf(p, trow[i], t);
        }
    }
    ss = wave_sum(s);
    dt = wave_sum(t);
}

template <bool BF16>
__global__ __launch_bounds__(256) void cosine_loss_fwd_kernel(
    const void *__restrict__ x, int64_t ldx, const int64_t *__restrict__ labels,
    const float *__restrict__ emb, int64_t lde, int64_t B, int64_t D, int64_t C,
    float *__restrict__ xhat, int64_t ldxhat, float *__restrict__ inv_norm, float *__restrict__ loss_i,
    int vec_ok)
{
    const int wave = threadIdx.x >> 6, lane = lane_id();
    for (int64_t row = (int64_t)blockIdx.x * LOSS_ROWS_PER_BLOCK + wave; row < B;
         row += (int64_t)gridDim.x * LOSS_ROWS_PER_BLOCK) {
        const char *xrow = (const char *)x + row * ldx * (BF16 ? 2 : 4);
        int64_t y = labels[row];
        y = y < 0 ? 0 : (y >= C ? C - 1 : y);  // clamp like a safe gather; host validates
        const float *trow = emb + y * lde;
        float ss, dt;
        row_reduce<BF16>(xrow, trow, D, vec_ok != 0, ss, dt);
        const float inv = 1.0f / sqrtf(fmaxf(ss, L2NORM_EPS));
        if (lane == 0) {
            if (inv_norm) inv_norm[row] = inv;
            loss_i[row] = 1.0f - dt * inv;
        }
        if (xhat) {
            float *orow = xhat + row * ldxhat;
            if (vec_ok && !BF16) {
                const float4 *xv = (const float4 *)xrow;
                float4 *ov = (float4 *)orow;
                for (int64_t i = lane; i < D / 4; i += WAVE) {
                    float4 p = xv[i];
                    ov[i] = make_float4(p.x * inv, p.y * inv, p.z * inv, p.w * inv);
                }
            } else {
                for (int64_t i = lane; i < D; i += WAVE) orow[i] = load_x<BF16>(xrow, i) * inv;
            }
        }
    }
}

// Deterministic mean of n floats: one 256-thread block, fixed tree.
__global__ __launch_bounds__(256) void mean_kernel(const float *__restrict__ v, int64_t n, float *__restrict__ out)
{
    __shared__ float part[256];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += v[i];
    part[threadIdx.x] = s;
    wg_barrier();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        wg_barrier();
    }
    if (threadIdx.x == 0) out[0] = part[0] / (float)n;
}

template <bool BF16, bool DX_BF16>
__global__ __launch_bounds__(256) void cosine_loss_bwd_kernel(
    const void *__restrict__ x, int64_t ldx, const int64_t *__restrict__ labels,
    const float *__restrict__ emb, int64_t lde, const float *__restrict__ grad_loss_i, float grad_scale,
    int64_t B, int64_t D, int64_t C, void *__restrict__ dx, int64_t lddx, int vec_ok)
{
    const int wave = threadIdx.x >> 6, lane = lane_id();
    for (int64_t row = (int64_t)blockIdx.x * LOSS_ROWS_PER_BLOCK + wave; row < B;
         row += (int64_t)gridDim.x * LOSS_ROWS_PER_BLOCK) {
        const char *xrow = (const char *)x + row * ldx * (BF16 ? 2 : 4);
        int64_t y = labels[row];
        y = y < 0 ? 0 : (y >= C ? C - 1 : y);
        const float *trow = emb + y * lde;
        float ss, dt;
        row_reduce<BF16>(xrow, trow, D, vec_ok != 0, ss, dt);
        const float w = grad_loss_i ? grad_loss_i[row] : grad_scale;
        const float inv = 1.0f / sqrtf(fmaxf(ss, L2NORM_EPS));
        const bool clamped = !(ss >= L2NORM_EPS);
        // g = -w t ; proj = xhat . g = -w inv (x . t)
        const float proj = clamped ? 0.f : (-w * inv * dt);
        // dx_d = (g_d - xhat_d proj) inv = (-w t_d - x_d inv proj) inv
        const float c_t = -w * inv;
        const float c_x = -inv * proj * inv;
        char *drow = (char *)dx + row * lddx * (DX_BF16 ? 2 : 4);
        for (int64_t i = lane; i < D; i += WAVE) {
            float v = fmaf(c_t, trow[i], c_x * load_x<BF16>(xrow, i));
            if constexpr (DX_BF16) ((uint16_t *)drow)[i] = f32_to_bf16(v);
            else ((float *)drow)[i] = v;
        }
    }
}

// ---- squared-distance loss of `--loss mse` (utils.squared_distance, utils.py:34-36; metric utils.mean_distance, :39-41) ----
// One wave per row, like the cosine head: loss_i = sum_d (x_d - E[y]_d)^2 (fp32 FMA accumulation per lane, wave tree), dist_i = sqrt(loss_i)
// when asked for; backward dx = 2 w (x - E[y]), w = grad_loss_i[row] or the scalar grad_scale.
template <bool BF16>
__global__ __launch_bounds__(256) void sqdist_loss_fwd_kernel(const void *__restrict__ x, int64_t ldx, const int64_t *__restrict__ labels,
                                                              const float *__restrict__ emb, int64_t lde, int64_t B, int64_t D, int64_t C,
                                                              float *__restrict__ loss_i, float *__restrict__ dist_i, int vec_ok)
{
    const int wave = threadIdx.x >> 6, lane = lane_id();
    for (int64_t row = (int64_t)blockIdx.x * LOSS_ROWS_PER_BLOCK + wave; row < B; row += (int64_t)gridDim.x * LOSS_ROWS_PER_BLOCK) {
        const char *xrow = (const char *)x + row * ldx * (BF16 ? 2 : 4);
        int64_t y = labels[row];
        y = y < 0 ? 0 : (y >= C ? C - 1 : y);
        const float *trow = emb + y * lde;
        float s = 0.f;
        if (vec_ok && !BF16) {
            const float4 *xv = (const float4 *)xrow, *tv = (const float4 *)trow;
            for (int64_t i = lane; i < D / 4; i += WAVE) {
                const float4 p = xv[i], q = tv[i];
                const float a = p.x - q.x, b = p.y - q.y, c = p.z - q.z, e = p.w - q.w;
                s = fmaf(a, a, s); s = fmaf(b, b, s); s = fmaf(c, c, s); s = fmaf(e, e, s);
            }
        } else {
            for (int64_t i = lane; i < D; i += WAVE) {
                const float a = load_x<BF16>(xrow, i) - trow[i];
                s = fmaf(a, a, s);
            }
        }
        s = wave_sum(s);
        if (lane == 0) {
            loss_i[row] = s;
            if (dist_i) dist_i[row] = sqrtf(s);
        }
    }
}

template <bool BF16, bool DX_BF16>
__global__ __launch_bounds__(256) void sqdist_loss_bwd_kernel(const void *__restrict__ x, int64_t ldx, const int64_t *__restrict__ labels,
                                                              const float *__restrict__ emb, int64_t lde, const float *__restrict__ grad_loss_i,
                                                              float grad_scale, int64_t B, int64_t D, int64_t C, void *__restrict__ dx, int64_t lddx)
{
    const int wave = threadIdx.x >> 6, lane = lane_id();
    for (int64_t row = (int64_t)blockIdx.x * LOSS_ROWS_PER_BLOCK + wave; row < B; row += (int64_t)gridDim.x * LOSS_ROWS_PER_BLOCK) {
        const char *xrow = (const char *)x + row * ldx * (BF16 ? 2 : 4);
        int64_t y = labels[row];
        y = y < 0 ? 0 : (y >= C ? C - 1 : y);
        const float *trow = emb + y * lde;
        const float w2 = 2.0f * (grad_loss_i ? grad_loss_i[row] : grad_scale);
        char *drow = (char *)dx + row * lddx * (DX_BF16 ? 2 : 4);
        for (int64_t i = lane; i < D; i += WAVE) {
            const float v = w2 * (load_x<BF16>(xrow, i) - trow[i]);
            if constexpr (DX_BF16) ((uint16_t *)drow)[i] = f32_to_bf16(v);
            else ((float *)drow)[i] = v;
        }
    }
}

// Stand-alone l2norm head (utils.py:125-127) for inference / feature dumps and for callers that
// keep the Keras-style ``Lambda(l2norm)`` layer separate from the loss.
template <bool BF16>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const void *__restrict__ x, int64_t ldx, int64_t B, int64_t D,
                                                         float *__restrict__ xhat, int64_t ldo, float *__restrict__ inv_norm)
{
    const int wave = threadIdx.x >> 6, lane = lane_id();
    for (int64_t row = (int64_t)blockIdx.x * LOSS_ROWS_PER_BLOCK + wave; row < B;
         row += (int64_t)gridDim.x * LOSS_ROWS_PER_BLOCK) {
        const char *xrow = (const char *)x + row * ldx * (BF16 ? 2 : 4);
        float s = 0.f;
        for (int64_t i = lane; i < D; i += WAVE) { const float p = load_x<BF16>(xrow, i); s = fmaf(p, p, s); }
        const float inv = 1.0f / sqrtf(fmaxf(wave_sum(s), L2NORM_EPS));
        if (lane == 0 && inv_norm) inv_norm[row] = inv;
        float *orow = xhat + row * ldo;
        for (int64_t i = lane; i < D; i += WAVE) orow[i] = load_x<BF16>(xrow, i) * inv;
    }
}

// dx = (g - xhat (xhat . g)) * inv_norm ; rows on the epsilon clamp (inv_norm == 1e6): dx = g * inv_norm
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float *__restrict__ g, int64_t ldg,
                                                         const float *__restrict__ xhat, int64_t ldh,
                                                         const float *__restrict__ inv_norm, int64_t B, int64_t D,
                                                         float *__restrict__ dx, int64_t lddx)
{
    const int wave = threadIdx.x >> 6, lane = lane_id();
    for (int64_t row = (int64_t)blockIdx.x * LOSS_ROWS_PER_BLOCK + wave; row < B;
         row += (int64_t)gridDim.x * LOSS_ROWS_PER_BLOCK) {
        const float *gr = g + row * ldg, *hr = xhat + row * ldh;
        float s = 0.f;
        for (int64_t i = lane; i < D; i += WAVE) s = fmaf(gr[i], hr[i], s);
        const float inv = inv_norm[row];
        const float proj = (inv >= 1e6f) ? 0.f : wave_sum(s);
        float *o = dx + row * lddx;
        for (int64_t i = lane; i < D; i += WAVE) o[i] = (gr[i] - hr[i] * proj) * inv;
    }
}

// ------------------------------------------------------------------------------------------------
// nn_accuracy: one wave owns a strip of 32 samples and walks all classes in tiles of 32 with
// v_mfma_f32_32x32x2_f32.  Operands are staged through LDS in K-chunks of 64, stored with even
// and odd k de-interleaved so that lanes 0-31 (k = 2t) and 32-63 (k = 2t+1) each read 16 B.
// Instead of materialising top-k, the metric uses the equivalent counting form
//     acc = (#within >= 1) && (#better < k)
// with  within = |score - true| < 1e-6,  better = score beyond true by >= 1e-6   (utils.py:84-95).
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ACC_BK = 64;            // k per staged chunk
constexpr int ACC_LD = ACC_BK + 4;    // padded row pitch (floats): conflict-free ds_read_b128

__device__ __forceinline__ void stage_tile32(float *lds, const float *src, int64_t ld, int64_t row0,
                                             int64_t nrows, int64_t k0, int64_t K)
{
    // 32 rows x 64 k, each lane moves 8 float4 (row = idx / 16, 4 consecutive k = idx % 16).  The loads are unconditional
    // (out-of-range elements read a valid address and are zeroed afterwards) so that all of them are in flight together.
    const int lane = lane_id();
    float v[8][4];
    const bool vec = ((ld | K) & 3) == 0 && (((uintptr_t)src) & 15) == 0;          // wave-uniform
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int idx = it * 64 + lane;
        const int r = idx >> 4, kq = (idx & 15) * 4;
        const bool rok = row0 + r < nrows;
        const float *p = src + (rok ? row0 + r : row0) * ld;
        if (vec) {
            const bool ok = rok && k0 + kq < K;
            const float4 t = *(const float4 *)(p + (k0 + kq < K ? k0 + kq : 0));
            v[it][0] = ok ? t.x : 0.f; v[it][1] = ok ? t.y : 0.f; v[it][2] = ok ? t.z : 0.f; v[it][3] = ok ? t.w : 0.f;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool kok = k0 + kq + j < K;
                const float t = p[kok ? k0 + kq + j : 0];
                v[it][j] = (rok && kok) ? t : 0.f;
            }
        }
    }
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int idx = it * 64 + lane;
        const int r = idx >> 4, kq = (idx & 15) * 4;
        float *o = lds + r * ACC_LD;
        // even k -> [0, 32), odd k -> [32, 64)
        o[(kq >> 1)] = v[it][0];
        o[(kq >> 1) + 1] = v[it][2];
        o[32 + (kq >> 1)] = v[it][1];
        o[32 + (kq >> 1) + 1] = v[it][3];
    }
}

__global__ __launch_bounds__(64) void nn_accuracy_kernel(
    const float *__restrict__ yp, int64_t ldp, const int64_t *__restrict__ labels,
    const float *__restrict__ emb, int64_t lde, int64_t B, int64_t D, int64_t C, int dot_prod_sim, int k,
    float *__restrict__ acc_out, float *__restrict__ scores, int64_t lds_, int32_t *__restrict__ best_out,
    int tiles_per_block, uint32_t *__restrict__ part_cnt, unsigned long long *__restrict__ part_best)
{
    // grid = (32-sample blocks, class slices of tiles_per_block x 32 classes).  One slice: results are written directly; several
    // (large class sets -- at ILSVRC size, C = D = 1000, a batch of 64 was TWO waves walking 32 class tiles each: 2.6 ms): every
    // slice adds its counts / proposes its best class into part_cnt / part_best and nn_accuracy_finish_kernel closes the metric.
    __shared__ __attribute__((aligned(16))) float sA[32 * ACC_LD];
    __shared__ __attribute__((aligned(16))) float sB[32 * ACC_LD];
    __shared__ float sTrue[32], sPn[32], sCn[32];

    const int lane = lane_id();
    const int col = lane & 31, hi = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 32;

    // "true" score per sample.  The reference forms it with a separate elementwise reduction
    // (utils.py:80 / :91) and then needs the 1e-6 band to absorb the float32 noise between that
    // reduction and the matmul.  Here the true score is rebuilt with exactly the arithmetic the
    // class-tile loop uses for column y (k-ascending fmaf chain == MFMA; |e|^2 as even-k sum +
    // odd-k sum), so the true class always matches its own matrix entry bit-for-bit.
    {
        const int64_t r = row0 + col;
        float t = 0.f, pn = 0.f;
        if (r < B) {
            int64_t y = labels[r];
            y = y < 0 ? 0 : (y >= C ? C - 1 : y);
            const float *p = yp + r * ldp, *e = emb + y * lde;
            if (hi == 0) {
                float s = 0.f;
                for (int64_t d = 0; d < D; d++) s = fmaf(p[d], e[d], s);
                t = s;                                   // S[r, y]
            } else if (!dot_prod_sim) {
                float ce = 0.f, co = 0.f;
                for (int64_t d = 0; d < D; d += 2) {
                    ce = fmaf(e[d], e[d], ce);
                    if (d + 1 < D) co = fmaf(e[d + 1], e[d + 1], co);
                }
                for (int64_t d = 0; d < D; d++) pn = fmaf(p[d], p[d], pn);
                t = ce + co;                             // |e_y|^2
            }
        }
        const float other = __shfl_xor(t, 32, 64);       // hi==0: |e_y|^2 ; hi==1: S[r, y]
        pn = __shfl_xor(pn, 32, 64);                     // hi==0 lanes receive |p|^2
        if (hi == 0) {
            sPn[col] = pn;
            sTrue[col] = dot_prod_sim ? t : ((pn + other) - 2.0f * t);
        }
    }
    wg_barrier();

    int n_better[16], n_within[16];
    float best_v[16];
    int best_c[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        n_better[r] = 0; n_within[r] = 0;
        best_v[r] = dot_prod_sim ? -INFINITY : INFINITY;
        best_c[r] = 0x7FFFFFFF;
    }

    const int64_t c_beg = (int64_t)blockIdx.y * tiles_per_block * 32;
    const int64_t c_end = (c_beg + (int64_t)tiles_per_block * 32 < C) ? c_beg + (int64_t)tiles_per_block * 32 : C;
    for (int64_t c0 = c_beg; c0 < c_end; c0 += 32) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        float cn = 0.f;  // |emb_c|^2 for this lane's class (Euclidean variant), half of D per lane
        for (int64_t k0 = 0; k0 < D; k0 += ACC_BK) {
            wg_barrier();
            stage_tile32(sA, yp, ldp, row0, B, k0, D);
            stage_tile32(sB, emb, lde, c0, C, k0, D);
            wg_barrier();
            const int64_t kc = (D - k0 < ACC_BK) ? (D - k0) : ACC_BK;
            const int steps = (int)((kc + 1) / 2);           // MFMA steps (2 k each), zero padded
            const float *pa = sA + col * ACC_LD + hi * 32;
            const float *pb = sB + col * ACC_LD + hi * 32;
            for (int s = 0; s < steps; s += 4) {
                const float4 a4 = *(const float4 *)(pa + s);
                const float4 b4 = *(const float4 *)(pb + s);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
                if (s + 1 < steps) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
                if (s + 2 < steps) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
                if (s + 3 < steps) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
                if (!dot_prod_sim) {
                    cn = fmaf(b4.x, b4.x, cn);
                    if (s + 1 < steps) cn = fmaf(b4.y, b4.y, cn);
                    if (s + 2 < steps) cn = fmaf(b4.z, b4.z, cn);
                    if (s + 3 < steps) cn = fmaf(b4.w, b4.w, cn);
                }
            }
        }
        if (!dot_prod_sim) {
            cn += __shfl_xor(cn, 32, 64);
            wg_barrier();
            if (hi == 0) sCn[col] = cn;
            wg_barrier();
        }
        const int64_t c = c0 + col;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;  // row of this accumulator register
            float sc = acc[r];
            if (!dot_prod_sim) sc = (sPn[lr] + sCn[col]) - 2.0f * sc;
            const bool valid = (c < C) && (row0 + lr < B);
            if (valid) {
                if (scores) scores[(row0 + lr) * lds_ + c] = sc;
                const float diff = dot_prod_sim ? (sc - sTrue[lr]) : (sTrue[lr] - sc);  // > 0: better than true
                if (fabsf(diff) < 1e-6f) n_within[r]++;
                else if (diff >= 1e-6f) n_better[r]++;
                const bool better = dot_prod_sim ? (sc > best_v[r]) : (sc < best_v[r]);
                if (better) { best_v[r] = sc; best_c[r] = (int)c; }
            }
        }
    }

    // reduce over the 32 lanes (classes) that share `hi`
#pragma unroll
    for (int r = 0; r < 16; r++) {
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            n_better[r] += __shfl_xor(n_better[r], off, 64);
            n_within[r] += __shfl_xor(n_within[r], off, 64);
            const float ov = __shfl_xor(best_v[r], off, 64);
            const int oc = __shfl_xor(best_c[r], off, 64);
            const bool take = dot_prod_sim ? (ov > best_v[r] || (ov == best_v[r] && oc < best_c[r]))
                                           : (ov < best_v[r] || (ov == best_v[r] && oc < best_c[r]));
            if (take) { best_v[r] = ov; best_c[r] = oc; }
        }
        const int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (col == 0 && row0 + lr < B) {
            if (gridDim.y == 1) {
                acc_out[row0 + lr] = (n_within[r] >= 1 && n_better[r] < k) ? 1.0f : 0.0f;
                if (best_out) best_out[row0 + lr] = best_c[r];
            } else {
                atomicAdd(&part_cnt[2 * (row0 + lr)], (uint32_t)n_better[r]);
                atomicAdd(&part_cnt[2 * (row0 + lr) + 1], (uint32_t)n_within[r]);
                if (best_c[r] != 0x7FFFFFFF) {
                    // best class of this slice as one ordered word: score (larger = better), then LOWER class index
                    float v = best_v[r];
                    if (v == 0.f) v = 0.f;                                     // -0 ties with +0
                    uint32_t u = __float_as_uint(v);
                    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);            // ascending with the value
                    if (!dot_prod_sim) u = ~u;                                 // distances: smaller is better
                    atomicMax(&part_best[row0 + lr], ((unsigned long long)u << 32) | (uint32_t)(0x7FFFFFFF - best_c[r]));
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void nn_accuracy_finish_kernel(const uint32_t *__restrict__ part_cnt, const unsigned long long *__restrict__ part_best,
                                                                int64_t B, int k, float *__restrict__ acc_out, int32_t *__restrict__ best_out)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= B) return;
    acc_out[r] = (part_cnt[2 * r + 1] >= 1u && part_cnt[2 * r] < (uint32_t)k) ? 1.0f : 0.0f;
    if (best_out) best_out[r] = 0x7FFFFFFF - (int32_t)(uint32_t)(part_best[r] & 0xFFFFFFFFull);
}

static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace se

using namespace se;

extern "C" int se_cosine_loss_fwd(const void *x, int x_dtype, int64_t ldx, const int64_t *labels,
                                  const float *emb, int64_t lde, int64_t B, int64_t D, int64_t C,
                                  float *xhat, int64_t ldxhat, float *inv_norm, float *loss_i,
                                  float *loss_mean, se_stream_t stream)
{
    if (B < 0 || D <= 0 || C <= 0) return fail(SE_ERR_INVALID, "se_cosine_loss_fwd: bad shape B=%lld D=%lld C=%lld", (long long)B, (long long)D, (long long)C);
    if (B == 0) return SE_OK;
    if (!x || !labels || !emb || !loss_i) return fail(SE_ERR_INVALID, "se_cosine_loss_fwd: null pointer");
    if (ldx < D || lde < D || (xhat && ldxhat < D)) return fail(SE_ERR_INVALID, "se_cosine_loss_fwd: leading dimension < D");
    if (x_dtype != SE_DTYPE_F32 && x_dtype != SE_DTYPE_BF16) return fail(SE_ERR_INVALID, "se_cosine_loss_fwd: bad dtype %d", x_dtype);
    hipStream_t s = (hipStream_t)stream;
    const bool bf = x_dtype == SE_DTYPE_BF16;
    const int vq = bf ? 8 : 4;
    const int vec_ok = (D % vq == 0) && (ldx % vq == 0) && (lde % 4 == 0) && aligned16(x) && aligned16(emb) &&
                       (!xhat || ((ldxhat % 4 == 0) && aligned16(xhat)));
    int64_t blocks = (B + LOSS_ROWS_PER_BLOCK - 1) / LOSS_ROWS_PER_BLOCK;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (bf) hipLaunchKernelGGL(cosine_loss_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, labels, emb, lde, B, D, C, xhat, ldxhat, inv_norm, loss_i, vec_ok);
    else hipLaunchKernelGGL(cosine_loss_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, labels, emb, lde, B, D, C, xhat, ldxhat, inv_norm, loss_i, vec_ok);
    SE_LAUNCH_CHECK();
    if (loss_mean) {
        hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, (const float *)loss_i, B, loss_mean);
        SE_LAUNCH_CHECK();
    }
    return SE_OK;
}

extern "C" int se_cosine_loss_bwd(const void *x, int x_dtype, int64_t ldx, const int64_t *labels,
                                  const float *emb, int64_t lde, const float *grad_loss_i, float grad_scale,
                                  int64_t B, int64_t D, int64_t C, void *dx, int dx_dtype, int64_t lddx,
                                  se_stream_t stream)
{
    if (B < 0 || D <= 0 || C <= 0) return fail(SE_ERR_INVALID, "se_cosine_loss_bwd: bad shape");
    if (B == 0) return SE_OK;
    if (!x || !labels || !emb || !dx) return fail(SE_ERR_INVALID, "se_cosine_loss_bwd: null pointer");
    if (ldx < D || lde < D || lddx < D) return fail(SE_ERR_INVALID, "se_cosine_loss_bwd: leading dimension < D");
    hipStream_t s = (hipStream_t)stream;
    const bool bf = x_dtype == SE_DTYPE_BF16, dbf = dx_dtype == SE_DTYPE_BF16;
    if ((x_dtype != SE_DTYPE_F32 && !bf) || (dx_dtype != SE_DTYPE_F32 && !dbf)) return fail(SE_ERR_INVALID, "se_cosine_loss_bwd: bad dtype");
    const int vq = bf ? 8 : 4;
    const int vec_ok = (D % vq == 0) && (ldx % vq == 0) && (lde % 4 == 0) && aligned16(x) && aligned16(emb);
    int64_t blocks = (B + LOSS_ROWS_PER_BLOCK - 1) / LOSS_ROWS_PER_BLOCK;
    if (blocks > 256 * 32) blocks = 256 * 32;
#define SE_BWD(XB, DB) hipLaunchKernelGGL((cosine_loss_bwd_kernel<XB, DB>), dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, labels, emb, lde, grad_loss_i, grad_scale, B, D, C, dx, lddx, vec_ok)
    if (bf && dbf) SE_BWD(true, true);
    else if (bf) SE_BWD(true, false);
    else if (dbf) SE_BWD(false, true);
    else SE_BWD(false, false);
#undef SE_BWD
    SE_LAUNCH_CHECK();
    return SE_OK;
}

extern "C" int se_sqdist_loss_fwd(const void *x, int x_dtype, int64_t ldx, const int64_t *labels, const float *emb, int64_t lde,
                                  int64_t B, int64_t D, int64_t C, float *loss_i, float *dist_i, float *loss_mean, se_stream_t stream)
{
    if (B < 0 || D <= 0 || C <= 0) return fail(SE_ERR_INVALID, "se_sqdist_loss_fwd: bad shape B=%lld D=%lld C=%lld", (long long)B, (long long)D, (long long)C);
    if (B == 0) return SE_OK;
    if (!x || !labels || !emb || !loss_i) return fail(SE_ERR_INVALID, "se_sqdist_loss_fwd: null pointer");
    if (ldx < D || lde < D) return fail(SE_ERR_INVALID, "se_sqdist_loss_fwd: leading dimension < D");
    if (x_dtype != SE_DTYPE_F32 && x_dtype != SE_DTYPE_BF16) return fail(SE_ERR_INVALID, "se_sqdist_loss_fwd: bad dtype %d", x_dtype);
    hipStream_t s = (hipStream_t)stream;
    const bool bf = x_dtype == SE_DTYPE_BF16;
    const int vec_ok = !bf && (D % 4 == 0) && (ldx % 4 == 0) && (lde % 4 == 0) && aligned16(x) && aligned16(emb);
    int64_t blocks = (B + LOSS_ROWS_PER_BLOCK - 1) / LOSS_ROWS_PER_BLOCK;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (bf) hipLaunchKernelGGL(sqdist_loss_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, labels, emb, lde, B, D, C, loss_i, dist_i, vec_ok);
    else hipLaunchKernelGGL(sqdist_loss_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, labels, emb, lde, B, D, C, loss_i, dist_i, vec_ok);
    SE_LAUNCH_CHECK();
    if (loss_mean) {
        hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, (const float *)loss_i, B, loss_mean);
        SE_LAUNCH_CHECK();
    }
    return SE_OK;
}

extern "C" int se_sqdist_loss_bwd(const void *x, int x_dtype, int64_t ldx, const int64_t *labels, const float *emb, int64_t lde,
                                  const float *grad_loss_i, float grad_scale, int64_t B, int64_t D, int64_t C, void *dx, int dx_dtype,
                                  int64_t lddx, se_stream_t stream)
{
    if (B < 0 || D <= 0 || C <= 0) return fail(SE_ERR_INVALID, "se_sqdist_loss_bwd: bad shape");
    if (B == 0) return SE_OK;
    if (!x || !labels || !emb || !dx) return fail(SE_ERR_INVALID, "se_sqdist_loss_bwd: null pointer");
    if (ldx < D || lde < D || lddx < D) return fail(SE_ERR_INVALID, "se_sqdist_loss_bwd: leading dimension < D");
    hipStream_t s = (hipStream_t)stream;
    const bool bf = x_dtype == SE_DTYPE_BF16, dbf = dx_dtype == SE_DTYPE_BF16;
    if ((x_dtype != SE_DTYPE_F32 && !bf) || (dx_dtype != SE_DTYPE_F32 && !dbf)) return fail(SE_ERR_INVALID, "se_sqdist_loss_bwd: bad dtype");
    int64_t blocks = (B + LOSS_ROWS_PER_BLOCK - 1) / LOSS_ROWS_PER_BLOCK;
    if (blocks > 256 * 32) blocks = 256 * 32;
#define SE_BWD(XB, DB) hipLaunchKernelGGL((sqdist_loss_bwd_kernel<XB, DB>), dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, labels, emb, lde, grad_loss_i, grad_scale, B, D, C, dx, lddx)
    if (bf && dbf) SE_BWD(true, true);
    else if (bf) SE_BWD(true, false);
    else if (dbf) SE_BWD(false, true);
    else SE_BWD(false, false);
#undef SE_BWD
    SE_LAUNCH_CHECK();
    return SE_OK;
}

// class tiles (of 32) one workgroup walks: all of them while that still fills the chip or the class set is small, else slices
static int nn_acc_tiles_per_block(int64_t B, int64_t C)
{
    const int64_t tiles = (C + 31) / 32, sample_blocks = (B + 31) / 32;
    if (tiles <= 4 || sample_blocks * 1 >= 1024) return (int)tiles;
    int64_t slices = 1024 / sample_blocks;                  // aim for ~1024 waves
    if (slices > tiles) slices = tiles;
    return (int)((tiles + slices - 1) / slices);
}

extern "C" int64_t se_nn_accuracy_workspace_bytes(int64_t B, int64_t C)
{
    if (B <= 0 || C <= 0) return 0;
    return nn_acc_tiles_per_block(B, C) * 32 >= C ? 0 : B * 16;      // per sample: two counters + one ordered word
}

extern "C" int se_nn_accuracy(const float *y_pred, int64_t ldp, const int64_t *labels, const float *emb,
                              int64_t lde, int64_t B, int64_t D, int64_t C, int dot_prod_sim, int k,
                              float *acc, float *scores, int64_t lds, int32_t *best, void *workspace, int64_t workspace_bytes,
                              se_stream_t stream)
{
    if (B < 0 || D <= 0 || C <= 0) return fail(SE_ERR_INVALID, "se_nn_accuracy: bad shape");
    if (B == 0) return SE_OK;
    if (!y_pred || !labels || !emb || !acc) return fail(SE_ERR_INVALID, "se_nn_accuracy: null pointer");
    if (ldp < D || lde < D || (scores && lds < C)) return fail(SE_ERR_INVALID, "se_nn_accuracy: leading dimension too small");
    if (k < 1) k = 1;
    const int64_t need = se_nn_accuracy_workspace_bytes(B, C);
    if (need > 0 && (!workspace || workspace_bytes < need || (((uintptr_t)workspace) & 7)))
        return fail(SE_ERR_WORKSPACE, "se_nn_accuracy: needs %lld bytes of 8-byte aligned workspace (se_nn_accuracy_workspace_bytes)", (long long)need);
    hipStream_t s = (hipStream_t)stream;
    const int tpb = nn_acc_tiles_per_block(B, C);
    const int64_t slices = ((C + 31) / 32 + tpb - 1) / tpb;
    unsigned long long *part_best = (unsigned long long *)workspace;
    uint32_t *part_cnt = need > 0 ? (uint32_t *)((char *)workspace + B * 8) : nullptr;
    if (need > 0) SE_HIP_CHECK(hipMemsetAsync(workspace, 0, (size_t)need, s));
    hipLaunchKernelGGL(nn_accuracy_kernel, dim3((unsigned)((B + 31) / 32), (unsigned)slices), dim3(64), 0, s,
                       y_pred, ldp, labels, emb, lde, B, D, C, dot_prod_sim, k, acc, scores, lds, best, tpb, part_cnt, part_best);
    SE_LAUNCH_CHECK();
    if (slices > 1) {
        hipLaunchKernelGGL(nn_accuracy_finish_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, part_cnt, part_best, B, k, acc, best);
        SE_LAUNCH_CHECK();
    }
    return SE_OK;
}

extern "C" int se_l2norm_fwd(const void *x, int x_dtype, int64_t ldx, int64_t B, int64_t D, float *xhat,
                             int64_t ldxhat, float *inv_norm, se_stream_t stream)
{
    if (B < 0 || D <= 0) return fail(SE_ERR_INVALID, "se_l2norm_fwd: bad shape");
    if (B == 0) return SE_OK;
    if (!x || !xhat || ldx < D || ldxhat < D) return fail(SE_ERR_INVALID, "se_l2norm_fwd: bad argument");
    int64_t blocks = (B + LOSS_ROWS_PER_BLOCK - 1) / LOSS_ROWS_PER_BLOCK;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (x_dtype == SE_DTYPE_BF16) hipLaunchKernelGGL(l2norm_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, B, D, xhat, ldxhat, inv_norm);
    else if (x_dtype == SE_DTYPE_F32) hipLaunchKernelGGL(l2norm_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, B, D, xhat, ldxhat, inv_norm);
    else return fail(SE_ERR_INVALID, "se_l2norm_fwd: bad dtype %d", x_dtype);
    SE_LAUNCH_CHECK();
    return SE_OK;
}

extern "C" int se_l2norm_bwd(const float *grad, int64_t ldg, const float *xhat, int64_t ldxhat, const float *inv_norm,
                             int64_t B, int64_t D, float *dx, int64_t lddx, se_stream_t stream)
{
    if (B < 0 || D <= 0) return fail(SE_ERR_INVALID, "se_l2norm_bwd: bad shape");
    if (B == 0) return SE_OK;
    if (!grad || !xhat || !inv_norm || !dx || ldg < D || ldxhat < D || lddx < D) return fail(SE_ERR_INVALID, "se_l2norm_bwd: bad argument");
    int64_t blocks = (B + LOSS_ROWS_PER_BLOCK - 1) / LOSS_ROWS_PER_BLOCK;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad, ldg, xhat, ldxhat, inv_norm, B, D, dx, lddx);
    SE_LAUNCH_CHECK();
    return SE_OK;
}
