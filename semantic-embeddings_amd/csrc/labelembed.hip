// labelembed.hip -- label-embedding baseline loss, forward + backward.
//
// Replaces `labelembed_loss(out1, out2, tar, targets, tau, alpha, beta)` (learn_labelembedding.py:21-37,
// with `cross_entropy` :17-18) and what TF autodiff derives from it.  Per sample i with label y:
//     L_o1_y   = -(log c_y - log sum_j c_j),  c = clip(softmax(out1), 1e-7, 1 - 1e-7)
//                (Keras 2.2 sparse_categorical_crossentropy on probabilities: clip, log, then TF's
//                 sparse_softmax_cross_entropy_with_logits, which renormalises the clipped probabilities)
//     L_o1_emb = -sum_c softmax(tar)_c * log_softmax(out1)_c          (softmax(tar) is stop_gradient)
//     L_o2_y   = the same on softmax(out2)
//     L_emb_o2 = -sum_c softmax(out2 / tau)_c * log_softmax(tar)_c * mask_i * B / (sum_j mask_j + 1e-8)
//                mask_i = [argmax(out2) == y]                         (softmax(out2/tau) and mask are stop_gradient)
//     L_re     = relu(softmax(out2)[y] - alpha)
//     loss_i   = beta L_o1_y + (1 - beta) L_o1_emb + L_o2_y + L_emb_o2 + L_re
//
// Layout: three [B, C] float32 logit matrices, one 64-lane wavefront per sample; every row statistic (3 maxima,
// 4 exp-sums, 2 weighted sums, the arg-max) is a DPP/shuffle wave reduction -- no LDS, no atomics.  The one
// batch-wide quantity, sum_j mask_j, is reduced in a fixed order by a single-block finish kernel, so results
// are run-to-run deterministic.  HBM-bound: 12 C bytes read per sample forward, 12 C read + 12 C written backward.
#include "se_common.h"

namespace se {

constexpr int LE_ROWS_PER_BLOCK = 4;   // 4 waves = 256 threads
constexpr float KERAS_EPS = 1e-7f;     // keras.backend.epsilon(): probabilities are clipped to [eps, 1 - eps]
constexpr int LE_AUX = 12;             // per-sample record kept for the backward pass

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// aux[i] = { lse(out1), lse(out2), lse(tar), lse(out2 / tau), mask, A_i = -sum tau2 * log_softmax(tar), base_i, p2_y,
//            S1 = sum_j clip(p1_j), R1 = sum_j 1{eps < p1_j < 1 - eps} p1_j, S2, R2 }
__global__ __launch_bounds__(256) void labelembed_fwd_kernel(const float *__restrict__ out1, int64_t ld1,
                                                             const float *__restrict__ out2, int64_t ld2,
                                                             const float *__restrict__ tar, int64_t ldt,
                                                             const int64_t *__restrict__ targets, int64_t B, int64_t C,
                                                             float tau, float alpha, float beta, float *__restrict__ aux)
{
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const float inv_tau = 1.0f / tau;
    for (int64_t row = (int64_t)blockIdx.x * LE_ROWS_PER_BLOCK + wave; row < B; row += (int64_t)gridDim.x * LE_ROWS_PER_BLOCK) {
        const float *o1 = out1 + row * ld1, *o2 = out2 + row * ld2, *tr = tar + row * ldt;
        int64_t y = targets[row];
        y = y < 0 ? 0 : (y >= C ? C - 1 : y);
        // pass 1: maxima and the arg-max of out2 (first maximum, like K.argmax)
        float m1 = -INFINITY, m2 = -INFINITY, mt = -INFINITY;
        int64_t am = 0;
        for (int64_t c = lane; c < C; c += WAVE) {
            m1 = fmaxf(m1, o1[c]);
            mt = fmaxf(mt, tr[c]);
            const float v = o2[c];
            if (v > m2) { m2 = v; am = c; }
        }
        m1 = wave_max(m1);
        mt = wave_max(mt);
        const float m2w = wave_max(m2);
        // lowest column holding the row maximum
        int64_t cand = (m2 == m2w) ? am : C;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int64_t o = __shfl_xor(cand, off, 64);
            cand = o < cand ? o : cand;
        }
        // pass 2: exp-sums and the two weighted sums
        float s1 = 0.f, s2 = 0.f, st = 0.f, s2t = 0.f, w_t_o1 = 0.f, w_tau_t = 0.f;
        for (int64_t c = lane; c < C; c += WAVE) {
            const float a = o1[c], b = o2[c], t = tr[c];
            s1 += expf(a - m1);
            s2 += expf(b - m2w);
            const float et = expf(t - mt), e2t = expf((b - m2w) * inv_tau);
            st += et;
            s2t += e2t;
            w_t_o1 += et * a;     // sum_c exp(t_c - mt) * out1_c
            w_tau_t += e2t * t;   // sum_c exp((o2_c - m2) / tau) * tar_c
        }
        s1 = wave_sum(s1); s2 = wave_sum(s2); st = wave_sum(st); s2t = wave_sum(s2t);
        w_t_o1 = wave_sum(w_t_o1); w_tau_t = wave_sum(w_tau_t);
        const float lse1 = m1 + logf(s1), lse2 = m2w + logf(s2);
        // pass 3: sums of the clipped probabilities (the renormalisation inside Keras' sparse CE) and of the un-clipped ones
        float S1 = 0.f, R1 = 0.f, S2 = 0.f, R2 = 0.f;
        for (int64_t c = lane; c < C; c += WAVE) {
            const float p1 = expf(o1[c] - lse1), p2 = expf(o2[c] - lse2);
            S1 += fminf(fmaxf(p1, KERAS_EPS), 1.0f - KERAS_EPS);
            S2 += fminf(fmaxf(p2, KERAS_EPS), 1.0f - KERAS_EPS);
            R1 += (p1 > KERAS_EPS && p1 < 1.0f - KERAS_EPS) ? p1 : 0.f;
            R2 += (p2 > KERAS_EPS && p2 < 1.0f - KERAS_EPS) ? p2 : 0.f;
        }
        S1 = wave_sum(S1); S2 = wave_sum(S2); R1 = wave_sum(R1); R2 = wave_sum(R2);
        if (lane == 0) {
            const float lset = mt + logf(st), lse2t = m2w * inv_tau + logf(s2t);
            const float p1y = expf(o1[y] - lse1), p2y = expf(o2[y] - lse2);
            const float l_o1_y = -(logf(fminf(fmaxf(p1y, KERAS_EPS), 1.0f - KERAS_EPS)) - logf(S1));
            const float l_o2_y = -(logf(fminf(fmaxf(p2y, KERAS_EPS), 1.0f - KERAS_EPS)) - logf(S2));
            const float l_o1_emb = -(w_t_o1 / st - lse1);         // -sum softmax(tar) * (out1 - lse1)
            const float a_i = -(w_tau_t / s2t - lset);            // -sum softmax(out2/tau) * (tar - lset)
            const float l_re = fmaxf(p2y - alpha, 0.f);
            float *r = aux + row * LE_AUX;
            r[0] = lse1; r[1] = lse2; r[2] = lset; r[3] = lse2t;
            r[4] = (cand == y) ? 1.0f : 0.0f;
            r[5] = a_i;
            r[6] = beta * l_o1_y + (1.0f - beta) * l_o1_emb + l_o2_y + l_re;
            r[7] = p2y;
            r[8] = S1; r[9] = R1; r[10] = S2; r[11] = R2;
        }
    }
}

// One block: sum of the masks in a fixed order -> scale = B / (sum + 1e-8); loss_i = base_i + A_i mask_i scale.
__global__ __launch_bounds__(256) void labelembed_finish_kernel(const float *__restrict__ aux, int64_t B, float *__restrict__ loss_i,
                                                                float *__restrict__ scale_out)
{
    __shared__ float part[256];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < B; i += 256) s += aux[i * LE_AUX + 4];
    part[threadIdx.x] = s;
    wg_barrier();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        wg_barrier();
    }
    const float scale = (float)B / (part[0] + 1e-8f);
    if (threadIdx.x == 0) scale_out[0] = scale;
    for (int64_t i = threadIdx.x; i < B; i += 256) loss_i[i] = aux[i * LE_AUX + 6] + aux[i * LE_AUX + 5] * aux[i * LE_AUX + 4] * scale;
}

// With in_j = 1{eps < p_j < 1 - eps} (clip_by_value passes gradient strictly inside the range), S = sum_j clip(p_j), R = sum_j in_j p_j:
//   ce'(p)_c = in_y (p_c - onehot_c) + p_c (in_c - R) / S          (= p - onehot when nothing is clipped)
// d out1 = g [ beta ce'(softmax1) + (1 - beta) (softmax1 - softmax(tar)) ]
// d out2 = g [ ce'(softmax2) + 1{p2_y > alpha} p2_y (onehot_y - softmax2) ]
// d tar  = g mask_i scale (softmax(tar) - softmax(out2 / tau))
__global__ __launch_bounds__(256) void labelembed_bwd_kernel(const float *__restrict__ out1, int64_t ld1, const float *__restrict__ out2,
                                                             int64_t ld2, const float *__restrict__ tar, int64_t ldt,
                                                             const int64_t *__restrict__ targets, const float *__restrict__ grad_loss_i,
                                                             float grad_scale, int64_t B, int64_t C, float tau, float alpha, float beta,
                                                             const float *__restrict__ aux, const float *__restrict__ scale,
                                                             float *__restrict__ d1, int64_t ldd1, float *__restrict__ d2, int64_t ldd2,
                                                             float *__restrict__ dt, int64_t lddt)
{
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const float inv_tau = 1.0f / tau, sc = scale[0];
    for (int64_t row = (int64_t)blockIdx.x * LE_ROWS_PER_BLOCK + wave; row < B; row += (int64_t)gridDim.x * LE_ROWS_PER_BLOCK) {
        const float *o1 = out1 + row * ld1, *o2 = out2 + row * ld2, *tr = tar + row * ldt;
        int64_t y = targets[row];
        y = y < 0 ? 0 : (y >= C ? C - 1 : y);
        const float *r = aux + row * LE_AUX;
        const float lse1 = r[0], lse2 = r[1], lset = r[2], lse2t = r[3], mask = r[4], p2y = r[7];
        const float g = grad_loss_i ? grad_loss_i[row] : grad_scale;
        const float p1y = expf(o1[y] - lse1);
        const float k1 = (p1y > KERAS_EPS && p1y < 1.0f - KERAS_EPS) ? 1.f : 0.f;   // clip_by_value passes gradient inside the range only
        const float k2 = (p2y > KERAS_EPS && p2y < 1.0f - KERAS_EPS) ? 1.f : 0.f;
        const float rS1 = 1.0f / r[8], R1 = r[9], rS2 = 1.0f / r[10], R2 = r[11];
        const float kre = (p2y > alpha) ? p2y : 0.f;
        const float wt = g * mask * sc;
        for (int64_t c = lane; c < C; c += WAVE) {
            const float sm1 = expf(o1[c] - lse1), sm2 = expf(o2[c] - lse2), smt = expf(tr[c] - lset);
            const float sm2t = expf(o2[c] * inv_tau - lse2t);
            const float hot = (c == y) ? 1.f : 0.f;
            const float in1 = (sm1 > KERAS_EPS && sm1 < 1.0f - KERAS_EPS) ? 1.f : 0.f, in2 = (sm2 > KERAS_EPS && sm2 < 1.0f - KERAS_EPS) ? 1.f : 0.f;
            if (d1) d1[row * ldd1 + c] = g * (beta * (k1 * (sm1 - hot) + sm1 * (in1 - R1) * rS1) + (1.0f - beta) * (sm1 - smt));
            if (d2) d2[row * ldd2 + c] = g * ((k2 * (sm2 - hot) + sm2 * (in2 - R2) * rS2) + kre * (hot - sm2));
            if (dt) dt[row * lddt + c] = wt * (smt - sm2t);
        }
    }
}

}  // namespace se

using namespace se;

static int le_grid(int64_t B)
{
    int64_t g = (B + LE_ROWS_PER_BLOCK - 1) / LE_ROWS_PER_BLOCK;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" int64_t se_labelembed_aux_floats(int64_t B) { return B > 0 ? B * LE_AUX + 4 : 0; }

extern "C" int se_labelembed_loss_fwd(const float *out1, int64_t ld1, const float *out2, int64_t ld2, const float *tar,
                                      int64_t ldt, const int64_t *targets, int64_t B, int64_t C, float tau, float alpha,
                                      float beta, float *loss_i, float *aux, se_stream_t stream)
{
    if (B < 0 || C <= 0) return fail(SE_ERR_INVALID, "se_labelembed_loss_fwd: bad shape B=%lld C=%lld", (long long)B, (long long)C);
    if (B == 0) return SE_OK;
    if (!out1 || !out2 || !tar || !targets || !loss_i || !aux) return fail(SE_ERR_INVALID, "se_labelembed_loss_fwd: null pointer");
    if (ld1 < C || ld2 < C || ldt < C) return fail(SE_ERR_INVALID, "se_labelembed_loss_fwd: leading dimension too small");
    if (!(tau > 0.f)) return fail(SE_ERR_INVALID, "se_labelembed_loss_fwd: tau must be positive");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(labelembed_fwd_kernel, dim3(le_grid(B)), dim3(256), 0, s, out1, ld1, out2, ld2, tar, ldt, targets, B, C, tau,
                       alpha, beta, aux);
    SE_LAUNCH_CHECK();
    hipLaunchKernelGGL(labelembed_finish_kernel, dim3(1), dim3(256), 0, s, aux, B, loss_i, aux + B * LE_AUX);
    SE_LAUNCH_CHECK();
    return SE_OK;
}

extern "C" int se_labelembed_loss_bwd(const float *out1, int64_t ld1, const float *out2, int64_t ld2, const float *tar,
                                      int64_t ldt, const int64_t *targets, const float *grad_loss_i, float grad_scale,
                                      int64_t B, int64_t C, float tau, float alpha, float beta, const float *aux,
                                      float *d_out1, int64_t ldd1, float *d_out2, int64_t ldd2, float *d_tar, int64_t lddt,
                                      se_stream_t stream)
{
    if (B < 0 || C <= 0) return fail(SE_ERR_INVALID, "se_labelembed_loss_bwd: bad shape B=%lld C=%lld", (long long)B, (long long)C);
    if (B == 0) return SE_OK;
    if (!out1 || !out2 || !tar || !targets || !aux) return fail(SE_ERR_INVALID, "se_labelembed_loss_bwd: null pointer");
    if ((d_out1 && ldd1 < C) || (d_out2 && ldd2 < C) || (d_tar && lddt < C)) return fail(SE_ERR_INVALID, "se_labelembed_loss_bwd: leading dimension too small");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(labelembed_bwd_kernel, dim3(le_grid(B)), dim3(256), 0, s, out1, ld1, out2, ld2, tar, ldt, targets, grad_loss_i,
                       grad_scale, B, C, tau, alpha, beta, aux, aux + B * LE_AUX, d_out1, ldd1, d_out2, ldd2, d_tar, lddt);
    SE_LAUNCH_CHECK();
    return SE_OK;
}
