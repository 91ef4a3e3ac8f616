// Pieces of the field split's set-up behind a Jacobian assembly that more than one kernel forms (and must form
// bit for bit alike): the inverse of a vertex's species block, and one entry of the half-precision species planes.
#pragma once
#include <hip/hip_runtime.h>

namespace fedm {

// inverse of the leading NS x NS (species) part of a diagonal block (Gauss-Jordan, partial pivoting)
template <int NS>
__device__ __forceinline__ void invert_species_block(double (&A)[NS][NS], double (&I)[NS][NS]) {
#pragma unroll
    for (int r = 0; r < NS; ++r)
#pragma unroll
        for (int cidx = 0; cidx < NS; ++cidx) I[r][cidx] = (r == cidx) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        int piv = k;
        double best = fabs(A[k][k]);
#pragma unroll
        for (int r = k + 1; r < NS; ++r)
            if (fabs(A[r][k]) > best) {
                best = fabs(A[r][k]);
                piv = r;
            }
#pragma unroll
        for (int r = k + 1; r < NS; ++r)
            if (piv == r) {
#pragma unroll
                for (int cidx = 0; cidx < NS; ++cidx) {
                    double t = A[k][cidx];
                    A[k][cidx] = A[r][cidx];
                    A[r][cidx] = t;
                    t = I[k][cidx];
                    I[k][cidx] = I[r][cidx];
                    I[r][cidx] = t;
                }
            }
        const double inv = 1.0 / A[k][k];
#pragma unroll
        for (int cidx = 0; cidx < NS; ++cidx) {
            A[k][cidx] *= inv;
            I[k][cidx] *= inv;
        }
#pragma unroll
        for (int r = 0; r < NS; ++r) {
            if (r == k) continue;
            const double f = A[r][k];
#pragma unroll
            for (int cidx = 0; cidx < NS; ++cidx) {
                A[r][cidx] -= f * A[k][cidx];
                I[r][cidx] -= f * I[k][cidx];
            }
        }
    }
}

// One (row, block column) entry of S = Duu^-1 J_uu in half precision.  zs: bit (r * NS + c) marks a species plane
// of the Jacobian that is structurally zero (J must hold 0 there).  Out-of-range entries saturate: only the
// preconditioner's quality is at stake.
template <int NS>
__device__ __forceinline__ void species_plane_entry(const double (&d)[NS][NS], const double (&J)[NS][NS], unsigned zs,
                                                    _Float16 (&row16)[NS * NS]) {
#pragma unroll
    for (int r = 0; r < NS; ++r)
#pragma unroll
        for (int cidx = 0; cidx < NS; ++cidx) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < NS; ++m) acc += d[r][m] * J[m][cidx];
            const float f = fminf(fmaxf((float)acc, -65504.f), 65504.f);
            row16[r * NS + cidx] = ((zs >> (r * NS + cidx)) & 1u) ? (_Float16)0.f : (_Float16)f;
        }
}

}  // namespace fedm
