// devise.hip -- DeViSE ranking loss on the class-embedding contraction, forward + backward.
//
// Replaces `utils.devise_ranking_loss(embedding, margin)(y_true, y_pred)` (utils.py:103-122):
//     true_sim_i  = sum_d y_true[i, d] * y_pred[i, d]
//     other_sim   = y_pred . E^T                                  [B, C]
//     loss_i      = sum_c relu(margin - true_sim_i + other_sim[i, c]) - margin
// and what TF autodiff derives from it w.r.t. y_pred (y_true is the target):
//     d y_pred[i, :] = g_i * ( sum_c a_ic * E[c, :]  -  n_i * y_true[i, :] ),   a_ic = [margin - true_sim_i + other_sim[i, c] > 0],
//                                                                               n_i = sum_c a_ic.
// Both contractions run on v_mfma_f32_32x32x2_f32 (one wave per 32 x 32 output tile, operands staged through LDS in K-chunks
// of 64 with even / odd k de-interleaved so every lane feeds four MFMA steps from one 16-byte LDS read -- the layout of
// se_nn_accuracy); the hinge, its row sums and the active mask are fused into the forward epilogue, the mask is the left
// operand of the backward contraction (0 / 1 entries: the sums of E rows are exact in chain order).
// y_true is either gathered on the device from the resident class embeddings (labels; learn_image_embeddings.py:48-50 feeds
// embedding[y]) or an explicit [B, D] matrix (the reference's calling convention).
#include "se_common.h"

namespace se {

typedef float dv_f32x16 __attribute__((ext_vector_type(16)));

constexpr int DV_BK = 64;            // k per staged chunk
constexpr int DV_LD = DV_BK + 4;     // padded row pitch (floats): conflict-free ds_read_b128
constexpr int DV_FWD_BK = 128;       // the forward kernel's chunk (one wave per workgroup: long chunks hide its global round trips)

// Staging loads are UNCONDITIONAL (out-of-range rows / k read a valid in-range address and are zeroed afterwards): with a branch per
// element the compiler waited for every load before issuing the next one, which was most of these kernels' time.
template <int BK = DV_BK>
__device__ __forceinline__ void dv_put(float *lds, int r, int kq, const float v[4])
{
    float *o = lds + r * (BK + 4);                   // even k to [0, BK / 2), odd k to [BK / 2, BK) of the row
    o[(kq >> 1)] = v[0];
    o[(kq >> 1) + 1] = v[2];
    o[BK / 2 + (kq >> 1)] = v[1];
    o[BK / 2 + (kq >> 1) + 1] = v[3];
}

// 8 x (one row's 4 consecutive k) per lane; rowp(r) = pointer to LDS row r's source row, or nullptr outside the matrix.
// Split in two so that a kernel can keep the NEXT chunk's loads in flight while the matrix pipe works on the current one.
template <int BK = DV_BK, class RowPtr>
__device__ __forceinline__ void dv_load_rows(float (&v)[BK / 8][4], RowPtr rowp, const float *any_valid_row, int64_t ld, int64_t k0, int64_t K)
{
    const int lane = lane_id();
    const bool vec = ((ld | K) & 3) == 0 && (((uintptr_t)any_valid_row) & 15) == 0;          // wave-uniform
#pragma unroll
    for (int it = 0; it < BK / 8; it++) {
        const int idx = it * 64 + lane;
        const int r = idx / (BK / 4), kq = (idx % (BK / 4)) * 4;
        const float *p = rowp(r);
        const bool rok = p != nullptr;
        if (!rok) p = any_valid_row;
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
}

template <int BK = DV_BK>
__device__ __forceinline__ void dv_put_rows(float *lds, const float (&v)[BK / 8][4])
{
    const int lane = lane_id();
#pragma unroll
    for (int it = 0; it < BK / 8; it++) {
        const int idx = it * 64 + lane;
        dv_put<BK>(lds, idx / (BK / 4), (idx % (BK / 4)) * 4, v[it]);
    }
}

template <class RowPtr>
__device__ __forceinline__ void dv_stage_rows(float *lds, RowPtr rowp, const float *any_valid_row, int64_t ld, int64_t k0, int64_t K)
{
    float v[DV_BK / 8][4];
    dv_load_rows<DV_BK>(v, rowp, any_valid_row, ld, k0, K);
    dv_put_rows<DV_BK>(lds, v);
}

// 32 rows x 64 k of a row-major [rows, K] matrix -> LDS; zero outside
__device__ __forceinline__ void dv_stage(float *lds, const float *src, int64_t ld, int64_t row0, int64_t nrows, int64_t k0, int64_t K)
{
    dv_stage_rows(lds, [=](int r) -> const float * { return row0 + r < nrows ? src + (row0 + r) * ld : nullptr; }, src, ld, k0, K);
}

// 32 TARGET rows x 64 k: row r is y_true[row0 + r] or, with labels, embedding[labels[row0 + r]] (clamped like the reference's gather)
__device__ __forceinline__ void dv_stage_target(float *lds, const float *yt, int64_t ldt, const int64_t *labels, const float *emb, int64_t lde,
                                                int64_t row0, int64_t B, int64_t C, int64_t k0, int64_t K)
{
    if (yt) {
        dv_stage(lds, yt, ldt, row0, B, k0, K);
        return;
    }
    dv_stage_rows(lds, [=](int r) -> const float * {
        if (row0 + r >= B) return nullptr;
        int64_t y = labels[row0 + r];
        y = y < 0 ? 0 : (y >= C ? C - 1 : y);
        return emb + y * lde;
    }, emb, lde, k0, K);
}

// the same tile of the TRANSPOSE of a row-major [K, cols] matrix: LDS row r = column col0 + r of `src`, k = its row index
__device__ __forceinline__ void dv_stage_t(float *lds, const float *src, int64_t ld, int64_t col0, int64_t ncols, int64_t k0, int64_t K)
{
    const int lane = lane_id();
    const int r = lane & 31;
    const bool cok = col0 + r < ncols;
    const float *p = src + (cok ? col0 + r : col0);
    float v[32];
#pragma unroll
    for (int it = 0; it < 32; it++) {
        const int k = it * 2 + (lane >> 5);                                 // 32 consecutive columns of one source row per half-wave
        const bool kok = k0 + k < K;
        const float t = p[(kok ? k0 + k : 0) * ld];
        v[it] = (cok && kok) ? t : 0.f;
    }
#pragma unroll
    for (int it = 0; it < 32; it++) {
        const int k = it * 2 + (lane >> 5);
        lds[r * DV_LD + ((k & 1) ? 32 : 0) + (k >> 1)] = v[it];
    }
}

template <int BK = DV_BK>
__device__ __forceinline__ dv_f32x16 dv_mma_chunk(dv_f32x16 acc, const float *sA, const float *sB, int col, int hi, int64_t kc)
{
    const int steps = (int)((kc + 1) / 2);
    const float *pa = sA + col * (BK + 4) + hi * (BK / 2);
    const float *pb = sB + col * (BK + 4) + hi * (BK / 2);
    for (int s = 0; s < steps; s += 4) {
        const float4 a4 = *(const float4 *)(pa + s);
        const float4 b4 = *(const float4 *)(pb + s);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
        if (s + 1 < steps) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
        if (s + 2 < steps) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
        if (s + 3 < steps) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
    }
    return acc;
}

// aux layout: true_sim [B] | n_active [B] | mask [B, C]

// true_sim_i = sum_d y_true[i, d] * y_pred[i, d] (utils.py:118): one wave per sample, coalesced loads, k-ascending fmaf chains per lane
// + a wave reduction.  Its own launch: inside devise_fwd_kernel it was a 16-chunk pre-pass in front of EVERY class slice (40 of the
// kernel's 128 us at C = D = 1000).
__global__ __launch_bounds__(256) void devise_true_sim_kernel(const float *__restrict__ yp, int64_t ldp, const int64_t *__restrict__ labels,
                                                             const float *__restrict__ yt, int64_t ldt, const float *__restrict__ emb, int64_t lde,
                                                             int64_t B, int64_t D, int64_t C, float *__restrict__ aux)
{
    const int lane = lane_id();
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B) return;
    const float *t;
    if (yt) t = yt + i * ldt;
    else {
        int64_t y = labels[i];
        y = y < 0 ? 0 : (y >= C ? C - 1 : y);
        t = emb + y * lde;
    }
    const float *p = yp + i * ldp;
    float acc = 0.f;
    for (int64_t k = lane; k < D; k += 64) acc = fmaf(p[k], t[k], acc);
    acc = wave_sum(acc);
    if (lane == 0) aux[i] = acc;
}

__global__ __launch_bounds__(64) void devise_fwd_kernel(const float *__restrict__ yp, int64_t ldp, const float *__restrict__ emb, int64_t lde,
                                                        int64_t B, int64_t D, int64_t C, float margin, float *__restrict__ loss_i,
                                                        float *__restrict__ aux, int tiles_per_block)
{
    // grid = (32-sample blocks, class slices of tiles_per_block x 32 classes).  One slice: results are written directly.  Several (large
    // class sets: one wave walking all C / 32 class tiles of its 32 samples was 4 waves on the chip for a batch of 128): every slice
    // leaves its partial hinge sums / active counts behind the mask in `aux` and devise_finish_kernel adds them in slice order
    // (deterministic: no floating-point atomics).
    // The (class tile, K-chunk) sequence is software-pipelined: while the MFMAs of one chunk run out of LDS the next chunk's operand
    // rows are already in flight into registers (one wave per workgroup: nothing else hides the round trip -- unpipelined, every one of
    // the 16 chunks of a D = 1000 tile exposed it).
    constexpr int FB = DV_FWD_BK;     // 128 k per chunk: 64 MFMA steps (1.7 us) per global round trip instead of 32
    __shared__ __attribute__((aligned(16))) float sA[32 * (FB + 4)];
    __shared__ __attribute__((aligned(16))) float sB[32 * (FB + 4)];
    __shared__ float sTrue[32];
    const int lane = lane_id();
    const int col = lane & 31, hi = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    float *mask = aux + 2 * B;
    if (lane < 32) sTrue[lane] = row0 + lane < B ? aux[row0 + lane] : 0.f;      // devise_true_sim_kernel ran ahead of this launch

    float hinge[16], nact[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { hinge[r] = 0.f; nact[r] = 0.f; }

    const int64_t c_beg = (int64_t)blockIdx.y * tiles_per_block * 32;
    const int64_t c_end = (c_beg + (int64_t)tiles_per_block * 32 < C) ? c_beg + (int64_t)tiles_per_block * 32 : C;
    const int64_t nchunks = (D + FB - 1) / FB;
    auto rows_a = [=](int r) -> const float * { return row0 + r < B ? yp + (row0 + r) * ldp : nullptr; };
    float va[FB / 8][4], vb[FB / 8][4];
    int64_t c0 = c_beg;
    if (c0 < c_end) {
        dv_load_rows<FB>(va, rows_a, yp, ldp, 0, D);
        dv_load_rows<FB>(vb, [=](int r) -> const float * { return c0 + r < C ? emb + (c0 + r) * lde : nullptr; }, emb, lde, 0, D);
    }
    dv_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    int64_t chunk = 0;
    while (c0 < c_end) {
        wg_barrier();                      // the previous chunk's MFMA reads of sA / sB are done
        dv_put_rows<FB>(sA, va);
        dv_put_rows<FB>(sB, vb);
        wg_barrier();
        // request the next chunk (same class tile, or the first chunk of the next one)
        const bool last_chunk = chunk + 1 == nchunks;
        const int64_t nc0 = last_chunk ? c0 + 32 : c0, nk0 = last_chunk ? 0 : (chunk + 1) * FB;
        if (nc0 < c_end) {
            dv_load_rows<FB>(va, rows_a, yp, ldp, nk0, D);
            dv_load_rows<FB>(vb, [=](int r) -> const float * { return nc0 + r < C ? emb + (nc0 + r) * lde : nullptr; }, emb, lde, nk0, D);
        }
        const int64_t k0 = chunk * FB;
        acc = dv_mma_chunk<FB>(acc, sA, sB, col, hi, (D - k0 < FB) ? (D - k0) : FB);
        if (last_chunk) {
            const int64_t c = c0 + col;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;      // row of this accumulator register
                const bool valid = (c < C) && (row0 + lr < B);
                const float h = (margin - sTrue[lr]) + acc[r];       // margin - true_sim[:, None] + other_sim (utils.py:120)
                const bool on = valid && h > 0.f;
                if (on) { hinge[r] += h; nact[r] += 1.f; }
                if (valid) mask[(row0 + lr) * C + c] = on ? 1.f : 0.f;
                acc[r] = 0.f;
            }
        }
        chunk = last_chunk ? 0 : chunk + 1;
        c0 = nc0;
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            hinge[r] += __shfl_xor(hinge[r], off, 64);
            nact[r] += __shfl_xor(nact[r], off, 64);
        }
        const int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (col == 0 && row0 + lr < B) {
            if (gridDim.y == 1) {
                loss_i[row0 + lr] = hinge[r] - margin;
                aux[B + row0 + lr] = nact[r];
            } else {
                float *part = aux + 2 * B + B * C + (int64_t)blockIdx.y * 2 * B;     // [slices][2][B]
                part[row0 + lr] = hinge[r];
                part[B + row0 + lr] = nact[r];
            }
        }
    }
}

__global__ __launch_bounds__(256) void devise_finish_kernel(float *__restrict__ aux, int64_t B, int64_t C, int slices, float margin,
                                                           float *__restrict__ loss_i)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= B) return;
    const float *part = aux + 2 * B + B * C;
    float h = 0.f, n = 0.f;
    for (int s = 0; s < slices; s++) { h += part[(int64_t)s * 2 * B + r]; n += part[(int64_t)s * 2 * B + B + r]; }
    loss_i[r] = h - margin;
    aux[B + r] = n;
}

static int devise_tiles_per_block(int64_t B, int64_t C)
{
    const int64_t tiles = (C + 31) / 32, sample_blocks = (B + 31) / 32;
    if (tiles <= 2 || sample_blocks >= 1024) return (int)tiles;
    int64_t slices = 1024 / sample_blocks;                  // aim for ~1024 waves
    if (slices > tiles) slices = tiles;
    return (int)((tiles + slices - 1) / slices);
}

constexpr int DV_BW = 4;      // waves per backward workgroup: each takes every 4th class chunk, sums meet in LDS in wave order

__global__ __launch_bounds__(64 * DV_BW) void devise_bwd_kernel(const int64_t *__restrict__ labels, const float *__restrict__ yt, int64_t ldt,
                                                                const float *__restrict__ emb, int64_t lde, const float *__restrict__ grad_loss_i,
                                                                float grad_scale, int64_t B, int64_t D, int64_t C, const float *__restrict__ aux,
                                                                float *__restrict__ dpred, int64_t lddp)
{
    // one 32 x 32 tile of d_pred per workgroup; a single wave walking all C / 64 chunks left a batch of 128 x D = 1000 on 128 waves
    // with 16 dependent stage -> MFMA rounds each, so the chunks are dealt to DV_BW waves with private staging buffers
    __shared__ __attribute__((aligned(16))) float sAB[DV_BW][2][32 * DV_LD];
    __shared__ float sRed[DV_BW - 1][16][64];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int col = lane & 31, hi = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 32, d0 = (int64_t)blockIdx.y * 32;
    const float *mask = aux + 2 * B;
    float *sA = sAB[wave][0], *sB = sAB[wave][1];
    dv_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    for (int64_t c0 = (int64_t)wave * DV_BK; c0 < C; c0 += DV_BW * DV_BK) {          // contraction over the classes
        dv_stage(sA, mask, C, row0, B, c0, C);                                        // wave-private buffers: a wave's LDS writes are
        dv_stage_t(sB, emb, lde, d0, D, c0, C);                                       // ordered against its own reads by the waitcnt
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        acc = dv_mma_chunk(acc, sA, sB, col, hi, (C - c0 < DV_BK) ? (C - c0) : DV_BK);
        __builtin_amdgcn_wave_barrier();
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) sRed[wave - 1][r][lane] = acc[r];
    }
    wg_barrier();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < DV_BW - 1; w++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] += sRed[w][r][lane];
    const int64_t d = d0 + col;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int64_t i = row0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (i < B && d < D) {
            float t;
            if (yt) t = yt[i * ldt + d];
            else {
                int64_t y = labels[i];
                y = y < 0 ? 0 : (y >= C ? C - 1 : y);
                t = emb[y * lde + d];
            }
            const float g = grad_loss_i ? grad_loss_i[i] : grad_scale;
            dpred[i * lddp + d] = g * (acc[r] - aux[B + i] * t);
        }
    }
}

}  // namespace se

using namespace se;

extern "C" int64_t se_devise_aux_floats(int64_t B, int64_t C)
{
    if (B <= 0 || C <= 0) return 0;
    const int tpb = devise_tiles_per_block(B, C);
    const int64_t slices = ((C + 31) / 32 + tpb - 1) / tpb;
    return 2 * B + B * C + (slices > 1 ? slices * 2 * B : 0);       // true_sim, active count, mask [B, C], per-slice partial sums
}

extern "C" int se_devise_loss_fwd(const float *y_pred, int64_t ldp, const int64_t *labels, const float *y_true, int64_t ldt,
                                  const float *emb, int64_t lde, int64_t B, int64_t D, int64_t C, float margin, float *loss_i,
                                  float *aux, se_stream_t stream)
{
    if (B < 0 || D <= 0 || C <= 0) return fail(SE_ERR_INVALID, "se_devise_loss_fwd: bad shape B=%lld D=%lld C=%lld", (long long)B, (long long)D, (long long)C);
    if (B == 0) return SE_OK;
    if (!y_pred || !emb || !loss_i || !aux || (!labels && !y_true)) return fail(SE_ERR_INVALID, "se_devise_loss_fwd: null pointer");
    if (ldp < D || lde < D || (y_true && ldt < D)) return fail(SE_ERR_INVALID, "se_devise_loss_fwd: leading dimension < D");
    const int tpb = devise_tiles_per_block(B, C);
    const int64_t slices = ((C + 31) / 32 + tpb - 1) / tpb;
    hipLaunchKernelGGL(devise_true_sim_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, y_pred, ldp, labels, y_true, ldt, emb, lde,
                       B, D, C, aux);
    SE_LAUNCH_CHECK();
    hipLaunchKernelGGL(devise_fwd_kernel, dim3((unsigned)((B + 31) / 32), (unsigned)slices), dim3(64), 0, (hipStream_t)stream, y_pred, ldp, emb, lde, B, D,
                       C, margin, loss_i, aux, tpb);
    SE_LAUNCH_CHECK();
    if (slices > 1) {
        hipLaunchKernelGGL(devise_finish_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, aux, B, C, (int)slices, margin, loss_i);
        SE_LAUNCH_CHECK();
    }
    return SE_OK;
}

extern "C" int se_devise_loss_bwd(const int64_t *labels, const float *y_true, int64_t ldt, const float *emb, int64_t lde,
                                  const float *grad_loss_i, float grad_scale, int64_t B, int64_t D, int64_t C, const float *aux,
                                  float *d_pred, int64_t lddp, se_stream_t stream)
{
    if (B < 0 || D <= 0 || C <= 0) return fail(SE_ERR_INVALID, "se_devise_loss_bwd: bad shape");
    if (B == 0) return SE_OK;
    if (!emb || !aux || !d_pred || (!labels && !y_true)) return fail(SE_ERR_INVALID, "se_devise_loss_bwd: null pointer");
    if (lde < D || lddp < D || (y_true && ldt < D)) return fail(SE_ERR_INVALID, "se_devise_loss_bwd: leading dimension < D");
    if ((D + 31) / 32 > 65535) return fail(SE_ERR_UNSUPPORTED, "se_devise_loss_bwd: D too large");
    hipLaunchKernelGGL(devise_bwd_kernel, dim3((unsigned)((B + 31) / 32), (unsigned)((D + 31) / 32)), dim3(64 * DV_BW), 0, (hipStream_t)stream, labels,
                       y_true, ldt, emb, lde, grad_loss_i, grad_scale, B, D, C, aux, d_pred, lddp);
    SE_LAUNCH_CHECK();
    return SE_OK;
}
