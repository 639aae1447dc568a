// Weighted Procrustes / Kabsch pose solve for gfx950: one workgroup per (pair, decoder layer) fuses
//   sigmoid(overlap logits) -> weight normalisation -> weighted centroids -> 3x3 covariance -> 3x3 SVD ->
//   det-corrected rotation + translation
// i.e. the pose assembly of /root/reference/src/models/regtr.py:185-203 and compute_rigid_transform of
// /root/reference/src/utils/se3_torch.py:108-154, which the reference runs as ~20 torch ops + torch.svd + a host
// assert per pair.  Correspondences in both directions are concatenated exactly as the reference does:
//   a = [src_kp ; tgt_corr]   b = [src_corr ; tgt_kp]   w = sigmoid([src_logit ; tgt_logit]).
// Reductions are float64 with a fixed tree (deterministic); the SVD is a one-sided Jacobi in float64 executed by
// one lane (9 numbers), singular values sorted descending so that "flip V[:,2]" (se3_torch.py:145-147) hits the
// smallest one as LAPACK's ordering does in the reference.
#include "common.h"

namespace {

constexpr int PT = 256;

__device__ __forceinline__ double block_sum(double v, double* sh)
{
    v = rg_wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if (rg_lane() == 0) sh[wave] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int w = 0; w < PT / RG_WAVE; w++) t += sh[w];
    return t;
}

// One-sided Jacobi SVD of a 3x3 matrix A (row-major): A = U diag(S) V^T, S descending.
__device__ void svd3(const double A[9], double U[9], double S[3], double V[9])
{
    double B[9];   // columns get rotated until mutually orthogonal: B = A V
    for (int i = 0; i < 9; i++) { B[i] = A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; i++) {
                    alpha += B[3 * i + p] * B[3 * i + p];
                    beta += B[3 * i + q] * B[3 * i + q];
                    gamma += B[3 * i + p] * B[3 * i + q];
                }
                if (gamma == 0.0) continue;
                off = fmax(off, fabs(gamma) / sqrt(fmax(alpha * beta, 1e-300)));
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 3; i++) {
                    const double bp = B[3 * i + p], bq = B[3 * i + q];
                    B[3 * i + p] = c * bp - s * bq; B[3 * i + q] = s * bp + c * bq;
                    const double vp = V[3 * i + p], vq = V[3 * i + q];
                    V[3 * i + p] = c * vp - s * vq; V[3 * i + q] = s * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    double n[3];
    for (int j = 0; j < 3; j++) n[j] = sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
    int ord[3] = {0, 1, 2};   // sort columns by singular value, descending
    for (int a = 0; a < 2; a++)
        for (int b = a + 1; b < 3; b++)
            if (n[ord[b]] > n[ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    double Vs[9];
    for (int j = 0; j < 3; j++) {
        S[j] = n[ord[j]];
        for (int i = 0; i < 3; i++) { Vs[3 * i + j] = V[3 * i + ord[j]]; U[3 * i + j] = B[3 * i + ord[j]]; }
    }
    for (int i = 0; i < 9; i++) V[i] = Vs[i];
    // normalise U; rebuild (near-)null directions so U stays orthonormal
    const double tiny = 1e-300 + 1e-14 * S[0];
    for (int j = 0; j < 2; j++) {
        if (S[j] > tiny) for (int i = 0; i < 3; i++) U[3 * i + j] /= S[j];
        else {   // any unit vector orthogonal to the previous columns
            double e[3] = {0, 0, 0};
            if (j == 0) e[0] = 1.0;
            else {
                const double u0[3] = {U[0], U[3], U[6]};
                int m = fabs(u0[0]) < fabs(u0[1]) ? (fabs(u0[0]) < fabs(u0[2]) ? 0 : 2) : (fabs(u0[1]) < fabs(u0[2]) ? 1 : 2);
                double a[3] = {0, 0, 0};
                a[m] = 1.0;
                const double d = a[0] * u0[0] + a[1] * u0[1] + a[2] * u0[2];
                double nn = 0;
                for (int i = 0; i < 3; i++) { e[i] = a[i] - d * u0[i]; nn += e[i] * e[i]; }
                nn = sqrt(nn);
                for (int i = 0; i < 3; i++) e[i] /= nn;
            }
            for (int i = 0; i < 3; i++) U[3 * i + j] = e[i];
        }
    }
    {
        // third column: u2 = +-(u0 x u1), sign agreeing with A v2 (so that S[2] >= 0)
        const double u0[3] = {U[0], U[3], U[6]}, u1[3] = {U[1], U[4], U[7]};
        double c[3] = {u0[1] * u1[2] - u0[2] * u1[1], u0[2] * u1[0] - u0[0] * u1[2], u0[0] * u1[1] - u0[1] * u1[0]};
        const double dot = c[0] * U[2] + c[1] * U[5] + c[2] * U[8];
        const double sg = dot < 0 ? -1.0 : 1.0;
        for (int i = 0; i < 3; i++) U[3 * i + 2] = sg * c[i];
    }
}

struct ProArgs {
    const float* kp;      // [N_total, 3] coarsest-level points, stacked [src clouds..., tgt clouds...]
    const float* corr;    // [L, N_total, 3] predicted corresponding coordinates
    const float* logit;   // [L, N_total]   overlap logits
    const int* seg_off;   // [2B + 1]
    float* pose;          // [L, B, 3, 4]
    int B, n_total;
    int* status;          // optional: REGTR_STATUS_NONFINITE_POSE is OR-ed in when a pose comes out non-finite
};

__global__ void __launch_bounds__(PT) k_procrustes(ProArgs g)
{
    __shared__ double sh[PT / RG_WAVE];
    const int b = blockIdx.x, l = blockIdx.y;
    const int s0 = g.seg_off[b], s1 = g.seg_off[b + 1], t0 = g.seg_off[g.B + b], t1 = g.seg_off[g.B + b + 1];
    const int ns = s1 - s0, n = ns + (t1 - t0);
    const float* corr = g.corr + (size_t)l * g.n_total * 3;
    const float* logit = g.logit + (size_t)l * g.n_total;

    auto fetch = [&](int i, float (&a)[3], float (&bb)[3], float& w) {
        const bool is_src = i < ns;
        const int row = is_src ? s0 + i : t0 + (i - ns);
        const float* kpp = g.kp + 3 * (size_t)row;
        const float* cp = corr + 3 * (size_t)row;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            a[d] = is_src ? kpp[d] : cp[d];      // regtr.py:187-190
            bb[d] = is_src ? cp[d] : kpp[d];
        }
        w = 1.0f / (1.0f + expf(-logit[row]));   // torch.sigmoid, regtr.py:191-194
    };

    // pass 1: sum of weights
    double wsum = 0;
    for (int i = threadIdx.x; i < n; i += PT) {
        float a[3], bb[3], w;
        fetch(i, a, bb, w);
        wsum += w;
    }
    wsum = block_sum(wsum, sh);
    const float denom = fmaxf((float)wsum, 1e-6f);               // se3_torch.py:127-128 (_EPS)

    // pass 2: weighted centroids
    double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
    for (int i = threadIdx.x; i < n; i += PT) {
        float a[3], bb[3], w;
        fetch(i, a, bb, w);
        const float wn = w / denom;
#pragma unroll
        for (int d = 0; d < 3; d++) { ca[d] += (double)(a[d] * wn); cb[d] += (double)(bb[d] * wn); }
    }
#pragma unroll
    for (int d = 0; d < 3; d++) { ca[d] = block_sum(ca[d], sh); cb[d] = block_sum(cb[d], sh); }
    float caf[3], cbf[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { caf[d] = (float)ca[d]; cbf[d] = (float)cb[d]; }

    // pass 3: cov = sum (a - ca)(b - cb)^T wn                    se3_torch.py:131-133
    double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += PT) {
        float a[3], bb[3], w;
        fetch(i, a, bb, w);
        const float wn = w / denom;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) cov[3 * r + c] += (double)((a[r] - caf[r]) * ((bb[c] - cbf[c]) * wn));
    }
#pragma unroll
    for (int e = 0; e < 9; e++) cov[e] = block_sum(cov[e], sh);

    if (threadIdx.x == 0) {
        double U[9], S[3], V[9];
        svd3(cov, U, S, V);
        // rot = V U^T ; if det <= 0 flip V[:, 2]                 se3_torch.py:143-148
        double R[9];
        auto vut = [&](double sgn) {
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++)
                    R[3 * r + c] = V[3 * r] * U[3 * c] + V[3 * r + 1] * U[3 * c + 1] + sgn * V[3 * r + 2] * U[3 * c + 2];
        };
        vut(1.0);
        const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                           R[2] * (R[3] * R[7] - R[4] * R[6]);
        if (!(det > 0)) vut(-1.0);
        float* P = g.pose + ((size_t)l * g.B + b) * 12;
        bool bad = false;
        for (int r = 0; r < 3; r++) {
            const double t = -(R[3 * r] * caf[0] + R[3 * r + 1] * caf[1] + R[3 * r + 2] * caf[2]) + cbf[r];   // :151
            P[4 * r] = (float)R[3 * r]; P[4 * r + 1] = (float)R[3 * r + 1]; P[4 * r + 2] = (float)R[3 * r + 2];
            P[4 * r + 3] = (float)t;
            bad = bad || !(fabs(R[3 * r]) <= 2.0) || !(fabs(R[3 * r + 1]) <= 2.0) || !(fabs(R[3 * r + 2]) <= 2.0) || !(fabs(t) < 1e30);
        }
        // a NaN / Inf anywhere upstream (an f16 pair operand out of range that a ReLU or max swallowed on the way, a non-finite input
        // coordinate) ends here: R|t is the sum over every token of the pair
        if (bad && g.status) atomicOr(g.status, REGTR_STATUS_NONFINITE_POSE);
    }
}

}  // namespace

extern "C" {

int regtr_abi_version(void) { return REGTR_ABI_VERSION; }

int regtr_weighted_procrustes(const float* kp, const float* corr, const float* logit, const int* seg_off, int n_pairs,
                              int n_total, int n_layers, float* pose, int* status, void* stream)
{
    if (!kp || !corr || !logit || !seg_off || !pose || n_pairs < 1 || n_layers < 1 || n_total < 0) return RG_ERR_ARG;
    ProArgs g{kp, corr, logit, seg_off, pose, n_pairs, n_total, status};
    k_procrustes<<<dim3(n_pairs, n_layers), PT, 0, (hipStream_t)stream>>>(g);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

}  // extern "C"
