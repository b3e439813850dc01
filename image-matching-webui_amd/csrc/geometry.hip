// Batched geometric verification on the device (the step AFTER the path).
//
// The reference verifies one pair at a time on the host: imcui/ui/utils.py:424-456 `proc_ransac_matches` -> cv2.findHomography /
// cv2.findFundamentalMat (default method USAC_MAGSAC, threshold 8 px, confidence 0.9999, 10000 iterations, :42-45), called twice per
// pair by `compute_geometry` (:532-610).  cv2's samplers and MAGSAC's sigma-consensus cannot be reproduced (cv2 is not even installable
// here), so this is an ADDITIONAL entry of the reference's `ransac_zoo` -- "HIP_RANSAC" -- not a re-implementation of cv2: a plain
// RANSAC whose every step is specified below, runs for B pairs at once, and has a CPU restatement (oracle/geometry.py) it is tested
// against (PARITY UNPINNED with respect to cv2; tested against the restatement and against ground-truth geometry).
//
//   1. hypothesis k of pair b samples m = 4 (homography) / 8 (fundamental) DISTINCT matches with a counter-based generator
//      (splitmix64 of (seed, b, k, j, attempt)): no sequential state, every hypothesis of every pair is generated in parallel;
//   2. minimal solver in float64 on Hartley-normalised points: homography = the 8 x 8 system with h33 = 1 (Gaussian elimination,
//      partial pivoting); fundamental = null vector of the 8 x 9 epipolar system (complete pivoting), rank 2 enforced by removing the
//      weakest right-singular direction (3 x 3 Jacobi); degenerate samples are marked invalid;
//   3. every hypothesis is scored on every match: inlier <=> squared error < threshold^2, error = forward transfer distance
//      |H x0 - x1| (what cv2's RANSAC measures) / Sampson distance for F;
//   4. the sequential RANSAC logic is then replayed over the counts exactly as a one-hypothesis-at-a-time loop would run it: a
//      hypothesis with more inliers than every earlier one becomes the best and shrinks the iteration bound
//      log(1 - confidence) / log(1 - w^m); hypotheses at or past the bound are ignored (so the result equals the sequential algorithm's);
//   5. local optimisation: normalised least-squares DLT over the inliers of the winner (9 x 9 normal matrix, Jacobi eigenvectors),
//      kept when it has at least as many inliers.
// Layout: pts0 / pts1 [B][N][2] float32 (pixels), counts [B]; model [B][9] float64 row-major (H scaled to h33 = 1, F to unit Frobenius
// norm with its largest entry positive), mask [B][N] uint8, info [B][4] int32 = (inliers, hypotheses consumed, index of the winner, ok).
#include <stdlib.h>

#include "common.h"
#include "imcui_hip.h"

#define GEO_HY 16  // hypotheses scored per workgroup pass

namespace {

__host__ __device__ inline unsigned long long geo_hash(unsigned long long seed, unsigned b, unsigned k, unsigned j, unsigned attempt) {
    unsigned long long x = seed * 0x9E3779B97F4A7C15ull + (((unsigned long long)b << 40) ^ ((unsigned long long)k << 8) ^ (unsigned long long)j) +
                           (unsigned long long)attempt * 0xD1B54A32D192ED03ull;
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

// T = [[s, 0, -s cx], [0, s, -s cy], [0, 0, 1]] of m points (interleaved x, y)
__device__ inline void hartley(const double* p, int m, double& s, double& cx, double& cy) {
    cx = cy = 0.0;
    for (int i = 0; i < m; ++i) {
        cx += p[2 * i];
        cy += p[2 * i + 1];
    }
    cx /= m;
    cy /= m;
    double d = 0.0;
    for (int i = 0; i < m; ++i) d += sqrt((p[2 * i] - cx) * (p[2 * i] - cx) + (p[2 * i + 1] - cy) * (p[2 * i + 1] - cy));
    d /= m;
    s = 1.4142135623730951 / (d > 1e-12 ? d : 1e-12);
}

// M (row-major 3x3) <- inv(T1) M T0 for a homography, T1^T M T0 for a fundamental matrix; T = (s, cx, cy)
__device__ inline void denorm_h(double* M, double s0, double cx0, double cy0, double s1, double cx1, double cy1) {
    double A[9];  // M T0
    for (int r = 0; r < 3; ++r) {
        A[3 * r + 0] = M[3 * r + 0] * s0;
        A[3 * r + 1] = M[3 * r + 1] * s0;
        A[3 * r + 2] = M[3 * r + 2] - s0 * (M[3 * r + 0] * cx0 + M[3 * r + 1] * cy0);
    }
    // inv(T1) = [[1/s1, 0, cx1], [0, 1/s1, cy1], [0, 0, 1]]
    for (int c = 0; c < 3; ++c) {
        M[c] = A[c] / s1 + cx1 * A[6 + c];
        M[3 + c] = A[3 + c] / s1 + cy1 * A[6 + c];
        M[6 + c] = A[6 + c];
    }
}
__device__ inline void denorm_f(double* M, double s0, double cx0, double cy0, double s1, double cx1, double cy1) {
    double A[9];
    for (int r = 0; r < 3; ++r) {
        A[3 * r + 0] = M[3 * r + 0] * s0;
        A[3 * r + 1] = M[3 * r + 1] * s0;
        A[3 * r + 2] = M[3 * r + 2] - s0 * (M[3 * r + 0] * cx0 + M[3 * r + 1] * cy0);
    }
    // T1^T = [[s1, 0, 0], [0, s1, 0], [-s1 cx1, -s1 cy1, 1]]
    for (int c = 0; c < 3; ++c) {
        M[c] = s1 * A[c];
        M[3 + c] = s1 * A[3 + c];
        M[6 + c] = A[6 + c] - s1 * (cx1 * A[c] + cy1 * A[3 + c]);
    }
}

// cyclic Jacobi on a symmetric n x n matrix (n <= 9): eigenvalues on the diagonal of A, eigenvectors in the columns of V
template <int n>
__device__ inline void jacobi(double* A, double* V) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {  // columns p, q of A
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {  // rows p, q of A
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
}

// F <- F with its weakest right-singular direction removed (rank 2)
__device__ inline void rank2(double* F) {
    double G[9], V[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) G[3 * i + j] = F[i] * F[j] + F[3 + i] * F[3 + j] + F[6 + i] * F[6 + j];
    jacobi<3>(G, V);
    int m = 0;
    for (int i = 1; i < 3; ++i)
        if (G[4 * i] < G[4 * m]) m = i;
    const double v0 = V[m], v1 = V[3 + m], v2 = V[6 + m];
    for (int r = 0; r < 3; ++r) {
        const double fv = F[3 * r] * v0 + F[3 * r + 1] * v1 + F[3 * r + 2] * v2;
        F[3 * r] -= fv * v0;
        F[3 * r + 1] -= fv * v1;
        F[3 * r + 2] -= fv * v2;
    }
}

// final scale: homography h33 = 1; fundamental unit Frobenius norm, largest-magnitude entry positive.  false when not finite / zero.
__device__ inline bool normalise_model(double* M, int geometry) {
    for (int i = 0; i < 9; ++i)
        if (!(fabs(M[i]) <= 1.7e308)) return false;
    if (geometry == 0) {
        if (fabs(M[8]) < 1e-12) return false;
        const double inv = 1.0 / M[8];
        for (int i = 0; i < 9; ++i) M[i] *= inv;
        M[8] = 1.0;
        return true;
    }
    double nn = 0.0;
    int big = 0;
    for (int i = 0; i < 9; ++i) {
        nn += M[i] * M[i];
        if (fabs(M[i]) > fabs(M[big])) big = i;
    }
    if (!(nn > 1e-300)) return false;
    const double inv = (M[big] < 0.0 ? -1.0 : 1.0) / sqrt(nn);
    for (int i = 0; i < 9; ++i) M[i] *= inv;
    return true;
}

// squared error of match (x, y) -> (u, v) under model M
__device__ __forceinline__ double model_err2(const double* M, int geometry, double x, double y, double u, double v) {
    const double a = M[0] * x + M[1] * y + M[2], b = M[3] * x + M[4] * y + M[5], c = M[6] * x + M[7] * y + M[8];
    if (geometry == 0) {
        if (fabs(c) < 1e-12) return 1e300;
        const double dx = a / c - u, dy = b / c - v;
        return dx * dx + dy * dy;
    }
    // Sampson: (x1^T F x0)^2 / ((F x0)_1^2 + (F x0)_2^2 + (F^T x1)_1^2 + (F^T x1)_2^2)
    const double e = u * a + v * b + c;
    const double ta = M[0] * u + M[3] * v + M[6], tb = M[1] * u + M[4] * v + M[7];
    const double den = a * a + b * b + ta * ta + tb * tb;
    return den > 1e-300 ? e * e / den : 1e300;
}

// ---- 1 + 2: hypotheses
__global__ __launch_bounds__(128) void geo_hyp_kernel(const float* __restrict__ p0, const float* __restrict__ p1, const int* __restrict__ counts, int N, int K,
                                                      int geometry, unsigned long long seed, double* __restrict__ models, int* __restrict__ valid) {
    const int k = blockIdx.x * 128 + threadIdx.x, b = blockIdx.y;
    if (k >= K) return;
    const int n = counts[b] < N ? counts[b] : N;
    const int m = geometry == 0 ? 4 : 8;
    double* out = models + ((size_t)b * K + k) * 9;
    int* ok = valid + (size_t)b * K + k;
    *ok = 0;
    if (n < m) return;
    int idx[8];
    for (int j = 0; j < m; ++j) {
        unsigned attempt = 0;
        for (;;) {
            int c = (int)(geo_hash(seed, (unsigned)b, (unsigned)k, (unsigned)j, attempt) % (unsigned long long)n);
            if (attempt >= 16) {  // (a pathological run of collisions: walk to the next free index)
                bool dup = true;
                while (dup) {
                    dup = false;
                    for (int t = 0; t < j; ++t) dup |= idx[t] == c;
                    if (dup) c = (c + 1) % n;
                }
            }
            bool dup = false;
            for (int t = 0; t < j; ++t) dup |= idx[t] == c;
            if (!dup) {
                idx[j] = c;
                break;
            }
            ++attempt;
        }
    }
    double a[16], c[16];
    for (int j = 0; j < m; ++j) {
        const size_t o = ((size_t)b * N + idx[j]) * 2;
        a[2 * j] = p0[o];
        a[2 * j + 1] = p0[o + 1];
        c[2 * j] = p1[o];
        c[2 * j + 1] = p1[o + 1];
    }
    double s0, cx0, cy0, s1, cx1, cy1;
    hartley(a, m, s0, cx0, cy0);
    hartley(c, m, s1, cx1, cy1);
    for (int j = 0; j < m; ++j) {
        a[2 * j] = (a[2 * j] - cx0) * s0;
        a[2 * j + 1] = (a[2 * j + 1] - cy0) * s0;
        c[2 * j] = (c[2 * j] - cx1) * s1;
        c[2 * j + 1] = (c[2 * j + 1] - cy1) * s1;
    }
    double M[9];
    if (geometry == 0) {
        double A[8][9];  // augmented system, unknowns h1..h8 (h9 = 1)
        for (int j = 0; j < 4; ++j) {
            const double x = a[2 * j], y = a[2 * j + 1], u = c[2 * j], v = c[2 * j + 1];
            double* r0 = A[2 * j];
            double* r1 = A[2 * j + 1];
            r0[0] = x; r0[1] = y; r0[2] = 1.0; r0[3] = 0.0; r0[4] = 0.0; r0[5] = 0.0; r0[6] = -u * x; r0[7] = -u * y; r0[8] = u;
            r1[0] = 0.0; r1[1] = 0.0; r1[2] = 0.0; r1[3] = x; r1[4] = y; r1[5] = 1.0; r1[6] = -v * x; r1[7] = -v * y; r1[8] = v;
        }
        for (int col = 0; col < 8; ++col) {
            int piv = col;
            for (int r = col + 1; r < 8; ++r)
                if (fabs(A[r][col]) > fabs(A[piv][col])) piv = r;
            if (fabs(A[piv][col]) < 1e-10) return;  // degenerate sample (three collinear points ...)
            if (piv != col)
                for (int t = 0; t < 9; ++t) {
                    const double tmp = A[col][t];
                    A[col][t] = A[piv][t];
                    A[piv][t] = tmp;
                }
            for (int r = col + 1; r < 8; ++r) {
                const double f = A[r][col] / A[col][col];
                for (int t = col; t < 9; ++t) A[r][t] -= f * A[col][t];
            }
        }
        for (int r = 7; r >= 0; --r) {
            double acc = A[r][8];
            for (int t = r + 1; t < 8; ++t) acc -= A[r][t] * M[t];
            M[r] = acc / A[r][r];
        }
        M[8] = 1.0;
        denorm_h(M, s0, cx0, cy0, s1, cx1, cy1);
    } else {
        double A[8][9];
        int perm[9];
        for (int t = 0; t < 9; ++t) perm[t] = t;
        for (int j = 0; j < 8; ++j) {
            const double x = a[2 * j], y = a[2 * j + 1], u = c[2 * j], v = c[2 * j + 1];
            double* r = A[j];
            r[0] = u * x; r[1] = u * y; r[2] = u; r[3] = v * x; r[4] = v * y; r[5] = v; r[6] = x; r[7] = y; r[8] = 1.0;
        }
        for (int col = 0; col < 8; ++col) {  // complete pivoting: the free column ends up last
            int pr = col, pc = col;
            for (int r = col; r < 8; ++r)
                for (int t = col; t < 9; ++t)
                    if (fabs(A[r][t]) > fabs(A[pr][pc])) {
                        pr = r;
                        pc = t;
                    }
            if (fabs(A[pr][pc]) < 1e-10) return;
            if (pr != col)
                for (int t = 0; t < 9; ++t) {
                    const double tmp = A[col][t];
                    A[col][t] = A[pr][t];
                    A[pr][t] = tmp;
                }
            if (pc != col) {
                for (int r = 0; r < 8; ++r) {
                    const double tmp = A[r][col];
                    A[r][col] = A[r][pc];
                    A[r][pc] = tmp;
                }
                const int tp = perm[col];
                perm[col] = perm[pc];
                perm[pc] = tp;
            }
            for (int r = col + 1; r < 8; ++r) {
                const double f = A[r][col] / A[col][col];
                for (int t = col; t < 9; ++t) A[r][t] -= f * A[col][t];
            }
        }
        double z[9];
        z[8] = 1.0;
        for (int r = 7; r >= 0; --r) {
            double acc = -A[r][8];
            for (int t = r + 1; t < 8; ++t) acc -= A[r][t] * z[t];
            z[r] = acc / A[r][r];
        }
        for (int t = 0; t < 9; ++t) M[perm[t]] = z[t];
        rank2(M);
        denorm_f(M, s0, cx0, cy0, s1, cx1, cy1);
    }
    if (!normalise_model(M, geometry)) return;
    for (int i = 0; i < 9; ++i) out[i] = M[i];
    *ok = 1;
}

// ---- 3: inlier counts of GEO_HY hypotheses per workgroup
__global__ __launch_bounds__(256) void geo_count_kernel(const float* __restrict__ p0, const float* __restrict__ p1, const int* __restrict__ counts, int N, int K,
                                                        int geometry, double thr2, const double* __restrict__ models, const int* __restrict__ valid,
                                                        int* __restrict__ cnt) {
    __shared__ double Ms[GEO_HY][9];
    __shared__ int vs[GEO_HY];
    __shared__ int part[4][GEO_HY];
    const int b = blockIdx.y, k0 = blockIdx.x * GEO_HY, tid = threadIdx.x;
    const int n = counts[b] < N ? counts[b] : N;
    if (tid < GEO_HY * 9) {
        const int h = tid / 9, e = tid - 9 * h;
        Ms[h][e] = (k0 + h < K) ? models[((size_t)b * K + k0 + h) * 9 + e] : 0.0;
    }
    if (tid < GEO_HY) vs[tid] = (k0 + tid < K) ? valid[(size_t)b * K + k0 + tid] : 0;
    __syncthreads();
    int c[GEO_HY];
#pragma unroll
    for (int h = 0; h < GEO_HY; ++h) c[h] = 0;
    for (int i = tid; i < n; i += 256) {
        const size_t o = ((size_t)b * N + i) * 2;
        const double x = p0[o], y = p0[o + 1], u = p1[o], v = p1[o + 1];
#pragma unroll
        for (int h = 0; h < GEO_HY; ++h) c[h] += model_err2(Ms[h], geometry, x, y, u, v) < thr2 ? 1 : 0;
    }
#pragma unroll
    for (int h = 0; h < GEO_HY; ++h) {
        const int s = wave_sum_i(c[h]);
        if ((tid & 63) == 0) part[tid >> 6][h] = s;
    }
    __syncthreads();
    if (tid < GEO_HY && k0 + tid < K) cnt[(size_t)b * K + k0 + tid] = vs[tid] ? part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid] : -1;
}

// ---- 4: the sequential logic over the counts (one thread per pair)
__global__ void geo_select_kernel(const int* __restrict__ counts, int N, int K, int m, double conf, const int* __restrict__ cnt,
                                  const double* __restrict__ models, double* __restrict__ best_model, int* __restrict__ info, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int n = counts[b] < N ? counts[b] : N;
    int best = 0, bestk = -1, needed = K, used = 0;
    const double lc = log(1.0 - (conf < 0.0 ? 0.0 : (conf > 0.999999999 ? 0.999999999 : conf)));
    for (int k = 0; k < K && k < needed; ++k) {
        used = k + 1;
        const int c = cnt[(size_t)b * K + k];
        if (c > best) {
            best = c;
            bestk = k;
            const double w = (double)c / (double)(n > 0 ? n : 1);
            double wm = 1.0;
            for (int t = 0; t < m; ++t) wm *= w;
            if (wm >= 1.0 - 1e-15)
                needed = 0;
            else if (wm > 1e-300) {
                // log1p: log(1.0 - wm) is 0 once wm < 2^-53 (w^8 for an inlier ratio below ~1 %), and lc / 0 = -inf used to read as
                // "no more hypotheses needed".  A non-negative denominator or a non-finite quotient means "no bound": keep K.
                const double den = log1p(-wm);
                const double it = den < 0.0 ? ceil(lc / den) : (double)K;
                needed = (it == it && it < (double)K) ? (int)(it < 0.0 ? 0.0 : it) : K;
            }
        }
    }
    info[4 * b + 0] = best;
    info[4 * b + 1] = used;
    info[4 * b + 2] = bestk;
    info[4 * b + 3] = (bestk >= 0 && best >= m) ? 1 : 0;
    for (int i = 0; i < 9; ++i) best_model[9 * b + i] = bestk >= 0 ? models[((size_t)b * K + bestk) * 9 + i] : 0.0;
}

// mask + count of a model per pair
__global__ __launch_bounds__(256) void geo_mask_kernel(const float* __restrict__ p0, const float* __restrict__ p1, const int* __restrict__ counts, int N, int geometry,
                                                       double thr2, const double* __restrict__ model, const int* __restrict__ okflag, unsigned char* __restrict__ mask,
                                                       int* __restrict__ ninl) {
    __shared__ int part[4];
    const int b = blockIdx.y, tid = threadIdx.x, i = blockIdx.x * 256 + tid;
    const int n = counts[b] < N ? counts[b] : N;
    int in = 0;
    if (i < N) {
        if (i < n && okflag[b]) {
            const size_t o = ((size_t)b * N + i) * 2;
            in = model_err2(model + 9 * b, geometry, p0[o], p0[o + 1], p1[o], p1[o + 1]) < thr2 ? 1 : 0;
        }
        mask[(size_t)b * N + i] = (unsigned char)in;
    }
    const int s = wave_sum_i(in);
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) atomicAdd(ninl + b, part[0] + part[1] + part[2] + part[3]);
}

// ---- 5: least-squares refit over the inliers (one workgroup per pair)
__device__ inline double block_sum(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void geo_refit_kernel(const float* __restrict__ p0, const float* __restrict__ p1, const int* __restrict__ counts, int N, int geometry,
                                                        const unsigned char* __restrict__ mask, const int* __restrict__ okflag, double* __restrict__ refit, int* __restrict__ refit_ok) {
    __shared__ double sh[4];
    __shared__ double ATA[81];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = counts[b] < N ? counts[b] : N;
    const int m = geometry == 0 ? 4 : 8;
    if (tid == 0) refit_ok[b] = 0;
    if (!okflag[b]) return;  // (uniform per block)
    // normalisation of the inlier sets
    double sx0 = 0, sy0 = 0, sx1 = 0, sy1 = 0, cnt = 0;
    for (int i = tid; i < n; i += 256)
        if (mask[(size_t)b * N + i]) {
            const size_t o = ((size_t)b * N + i) * 2;
            sx0 += p0[o]; sy0 += p0[o + 1]; sx1 += p1[o]; sy1 += p1[o + 1]; cnt += 1.0;
        }
    cnt = block_sum(cnt, sh);
    if (cnt < (double)m) return;
    const double cx0 = block_sum(sx0, sh) / cnt, cy0 = block_sum(sy0, sh) / cnt, cx1 = block_sum(sx1, sh) / cnt, cy1 = block_sum(sy1, sh) / cnt;
    double d0 = 0, d1 = 0;
    for (int i = tid; i < n; i += 256)
        if (mask[(size_t)b * N + i]) {
            const size_t o = ((size_t)b * N + i) * 2;
            d0 += sqrt((p0[o] - cx0) * (p0[o] - cx0) + (p0[o + 1] - cy0) * (p0[o + 1] - cy0));
            d1 += sqrt((p1[o] - cx1) * (p1[o] - cx1) + (p1[o + 1] - cy1) * (p1[o + 1] - cy1));
        }
    d0 = block_sum(d0, sh) / cnt;
    d1 = block_sum(d1, sh) / cnt;
    const double s0 = 1.4142135623730951 / (d0 > 1e-12 ? d0 : 1e-12), s1 = 1.4142135623730951 / (d1 > 1e-12 ? d1 : 1e-12);
    // normal matrix A^T A (upper triangle, 45 sums)
    double acc[45];
#pragma unroll
    for (int t = 0; t < 45; ++t) acc[t] = 0.0;
    for (int i = tid; i < n; i += 256)
        if (mask[(size_t)b * N + i]) {
            const size_t o = ((size_t)b * N + i) * 2;
            const double x = (p0[o] - cx0) * s0, y = (p0[o + 1] - cy0) * s0, u = (p1[o] - cx1) * s1, v = (p1[o + 1] - cy1) * s1;
            if (geometry == 0) {
                const double r0[9] = {-x, -y, -1.0, 0.0, 0.0, 0.0, u * x, u * y, u};
                const double r1[9] = {0.0, 0.0, 0.0, -x, -y, -1.0, v * x, v * y, v};
                int t = 0;
#pragma unroll
                for (int i2 = 0; i2 < 9; ++i2)
#pragma unroll
                    for (int j2 = i2; j2 < 9; ++j2) acc[t++] += r0[i2] * r0[j2] + r1[i2] * r1[j2];
            } else {
                const double r[9] = {u * x, u * y, u, v * x, v * y, v, x, y, 1.0};
                int t = 0;
#pragma unroll
                for (int i2 = 0; i2 < 9; ++i2)
#pragma unroll
                    for (int j2 = i2; j2 < 9; ++j2) acc[t++] += r[i2] * r[j2];
            }
        }
    {
        int t = 0;
        for (int i2 = 0; i2 < 9; ++i2)
            for (int j2 = i2; j2 < 9; ++j2) {
                const double s = block_sum(acc[t++], sh);
                if (tid == 0) ATA[9 * i2 + j2] = ATA[9 * j2 + i2] = s;
            }
    }
    __syncthreads();
    if (tid != 0) return;
    double A[81], V[81];
    for (int i = 0; i < 81; ++i) A[i] = ATA[i];
    jacobi<9>(A, V);
    int mn = 0;
    for (int i = 1; i < 9; ++i)
        if (A[10 * i] < A[10 * mn]) mn = i;
    double M[9];
    for (int i = 0; i < 9; ++i) M[i] = V[9 * i + mn];
    if (geometry == 0) {
        denorm_h(M, s0, cx0, cy0, s1, cx1, cy1);
    } else {
        rank2(M);
        denorm_f(M, s0, cx0, cy0, s1, cx1, cy1);
    }
    if (!normalise_model(M, geometry)) return;
    for (int i = 0; i < 9; ++i) refit[9 * b + i] = M[i];
    refit_ok[b] = 1;
}

// keep the refit when it has at least as many inliers
__global__ void geo_choose_kernel(int N, double* __restrict__ model, unsigned char* __restrict__ mask, int* __restrict__ info, const double* __restrict__ refit,
                                  const unsigned char* __restrict__ mask2, const int* __restrict__ ninl2, const int* __restrict__ refit_ok, const int* __restrict__ ninl1) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool take = info[4 * b + 3] && refit_ok[b] && ninl2[b] >= ninl1[b];
    if (take) {
        for (int i = tid; i < N; i += blockDim.x) mask[(size_t)b * N + i] = mask2[(size_t)b * N + i];
        if (tid < 9) model[9 * b + tid] = refit[9 * b + tid];
    }
    __syncthreads();
    if (tid == 0) info[4 * b + 0] = info[4 * b + 3] ? (take ? ninl2[b] : ninl1[b]) : 0;
}

struct GeoWs {
    double *models, *refit;
    int *valid, *cnt, *ninl1, *ninl2, *refit_ok, *okflag;
    unsigned char* mask2;
};
GeoWs geo_carve(WsAlloc& a, int B, int N, int K) {
    GeoWs w;
    w.models = a.get<double>((size_t)B * K * 9);
    w.refit = a.get<double>((size_t)B * 9);
    w.valid = a.get<int>((size_t)B * K);
    w.cnt = a.get<int>((size_t)B * K);
    w.ninl1 = a.get<int>(B);
    w.ninl2 = a.get<int>(B);
    w.refit_ok = a.get<int>(B);
    w.okflag = a.get<int>(B);
    w.mask2 = a.get<unsigned char>((size_t)B * N);
    return w;
}

__global__ void geo_okflag_kernel(const int* __restrict__ info, int* __restrict__ ok, int* __restrict__ z0, int* __restrict__ z1, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        ok[b] = info[4 * b + 3];
        z0[b] = 0;
        z1[b] = 0;
    }
}

// between two local-optimisation rounds: the chosen model's inlier count becomes the count to beat, the refit's counter starts from zero
__global__ void geo_next_round_kernel(const int* __restrict__ info, int* __restrict__ ninl1, int* __restrict__ ninl2, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        ninl1[b] = info[4 * b + 0];
        ninl2[b] = 0;
    }
}

}  // namespace

#define GEO_MAX_HYP 16384
// least-squares refits on the current inliers, each kept when it has at least as many inliers as the model it came from (round 5: three rounds;
// the single refit of round 4 measured a median corner error of 7.7 px where a textbook RANSAC with two refits reached 4.6 px on the same matches,
// tests/test_gpu_real_images.py)
#define GEO_REFIT_ROUNDS 3

extern "C" size_t imcui_hip_ransac_workspace_bytes(int B, int N, int max_iter) {
    if (B <= 0 || N <= 0 || max_iter <= 0) return 0;
    const int K = max_iter < GEO_MAX_HYP ? max_iter : GEO_MAX_HYP;
    WsAlloc a(nullptr, 0);
    (void)geo_carve(a, B, N, K);
    return a.off + 256;
}

extern "C" int imcui_hip_ransac(imcui_hip_t* h, const float* pts0, const float* pts1, const int* counts, int B, int N, int geometry, double reproj_threshold,
                                double confidence, int max_iter, unsigned long long seed, double* model, unsigned char* mask, int* info, void* ws,
                                size_t ws_bytes, void* stream_) {
    if (!h || !pts0 || !pts1 || !counts || !model || !mask || !info) return imcui_set_err(h, IMCUI_ERR_ARG, "ransac: null argument");
    if (B <= 0 || N <= 0 || max_iter <= 0 || (geometry != 0 && geometry != 1) || !(reproj_threshold > 0.0))
        return imcui_set_err(h, IMCUI_ERR_ARG, "ransac: B=%d N=%d max_iter=%d geometry=%d threshold=%g", B, N, max_iter, geometry, reproj_threshold);
    if (!ws || ws_bytes < imcui_hip_ransac_workspace_bytes(B, N, max_iter)) return imcui_set_err(h, IMCUI_ERR_WS, "ransac: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const int K = max_iter < GEO_MAX_HYP ? max_iter : GEO_MAX_HYP;
    const int m = geometry == 0 ? 4 : 8;
    const double thr2 = reproj_threshold * reproj_threshold;
    WsAlloc a(ws, ws_bytes);
    GeoWs w = geo_carve(a, B, N, K);
    hipLaunchKernelGGL(geo_hyp_kernel, dim3((K + 127) / 128, B), dim3(128), 0, stream, pts0, pts1, counts, N, K, geometry, seed, w.models, w.valid);
    hipLaunchKernelGGL(geo_count_kernel, dim3((K + GEO_HY - 1) / GEO_HY, B), dim3(256), 0, stream, pts0, pts1, counts, N, K, geometry, thr2, w.models, w.valid, w.cnt);
    hipLaunchKernelGGL(geo_select_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, counts, N, K, m, confidence, w.cnt, w.models, model, info, B);
    hipLaunchKernelGGL(geo_okflag_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, info, w.okflag, w.ninl1, w.ninl2, B);
    hipLaunchKernelGGL(geo_mask_kernel, dim3((N + 255) / 256, B), dim3(256), 0, stream, pts0, pts1, counts, N, geometry, thr2, model, w.okflag, mask, w.ninl1);
    for (int round = 0; round < GEO_REFIT_ROUNDS; ++round) {
        if (round > 0) hipLaunchKernelGGL(geo_next_round_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, info, w.ninl1, w.ninl2, B);
        hipLaunchKernelGGL(geo_refit_kernel, dim3(B), dim3(256), 0, stream, pts0, pts1, counts, N, geometry, mask, w.okflag, w.refit, w.refit_ok);
        hipLaunchKernelGGL(geo_mask_kernel, dim3((N + 255) / 256, B), dim3(256), 0, stream, pts0, pts1, counts, N, geometry, thr2, w.refit, w.refit_ok, w.mask2, w.ninl2);
        hipLaunchKernelGGL(geo_choose_kernel, dim3(B), dim3(256), 0, stream, N, model, mask, info, w.refit, w.mask2, w.ninl2, w.refit_ok, w.ninl1);
    }
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}
