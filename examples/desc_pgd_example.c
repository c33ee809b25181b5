/* Minimal C client of the C ABI (include/desc_amd.h): what the MEX shim does, without MATLAB.
 *
 *   gcc -std=c99 -I include examples/desc_pgd_example.c -L desc_amd -ldesc_amd -lm -o /tmp/desc_example
 *   LD_LIBRARY_PATH=desc_amd /tmp/desc_example
 *
 * Builds the complete graph on 8 nodes with exact relative rotations R_ij = R_i R_j' about the z axis
 * except for one corrupted edge, runs DESC_PGD (Algorithms/DESC_PGD.m:14) for up to 100 iterations with
 * ConstantStepSize(0.01) (the patience rule of :243-256 may stop earlier) and prints s_ij: ~0 on the consistent
 * edges, the rotation error / pi on the corrupted one. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "desc_amd.h"

static void rotz(double a, double* R) {          /* column-major 3x3, like RijMat(:,:,l) */
    R[0] = cos(a); R[1] = sin(a); R[2] = 0; R[3] = -sin(a); R[4] = cos(a); R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
}

int main(void) {
    enum { N = 8, M = N * (N - 1) / 2 };
    static int32_t ii[M], jj[M];
    static double ind[2 * M];                   /* Ind as MATLAB holds it: M x 2 doubles, column-major, 1-based, sorted by (i, j) (DESC_PGD.m:5) */
    static double rij[9 * M], s_vec[M], obj[100], avg[100];
    int e = 0, bad = 5;
    for (int i = 0; i < N; ++i)
        for (int j = i + 1; j < N; ++j, ++e) {
            ind[e] = i + 1; ind[M + e] = j + 1;
            rotz(0.3 * i - 0.3 * j + (e == bad ? 1.0 : 0.0), rij + 9 * e);       /* R_i R_j' = rotz(a_i - a_j) */
        }
    int64_t n = 0; int32_t sorted = 0;          /* -> 0-based int32 endpoints, n = max(Ind(:)), checked; column-major M x 2: strides (1, M) */
    if (desc_marshal_edges(ind, DESC_DTYPE_F64, M, 1, M, ii, jj, &n, &sorted) != DESC_OK || !sorted || n != N) {
        fprintf(stderr, "desc_marshal_edges: %s\n", desc_last_error()); return 1;
    }
    desc_problem prob = {n, M, ii, jj, rij};
    desc_params p;
    desc_params_default(&p);                    /* iters 100, ConstantStepSize(0.01), patience 30 */
    desc_result r = {0};
    r.s_vec = s_vec; r.obj_trace = obj; r.avg_change_trace = avg;
    int rc = desc_pgd_solve(&prob, &p, &r);
    if (rc != DESC_OK) { fprintf(stderr, "desc_pgd_solve: %d: %s\n", rc, desc_last_error()); return 1; }
    printf("%s: %d iterations, %.2f ms\n", desc_version(), (int)r.iters_run, r.ms_total);
    double worst_good = 0;
    for (e = 0; e < M; ++e) if (e != bad && s_vec[e] > worst_good) worst_good = s_vec[e];
    printf("s(corrupted edge %d-%d) = %.4f, max s over the consistent edges = %.4f\n", ii[bad], jj[bad], s_vec[bad], worst_good);
    return s_vec[bad] > 2 * worst_good ? 0 : 2;
}
