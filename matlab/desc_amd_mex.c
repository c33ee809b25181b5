/* desc_amd_mex.c -- MEX dispatcher for the "next" rows of the C ABI (include/desc_amd.h):
 *
 *   R     = desc_amd_mex('spectral', Ind0, RijMat)                       desc_spectral_run, Spectral.m
 *   R     = desc_amd_mex('gcw',      Ind0, RijMat, S_vec)                desc_spectral_run, GCW.m
 *   SVec  = desc_amd_mex('cemp',     Ind0, RijMat, beta, max_iter, nsample, seed)   desc_cemp_run, CEMP.m
 *   R     = desc_amd_mex('refine',   Ind0, RijMat, S_vec, R_init)        desc_refine_run, DESC.m:265-313
 *
 * Ind0: m x 2 int32, 0-based, sorted by (i,j); RijMat: 3 x 3 x m double (passed through);
 * R: 3 x 3 x n double.  Cannot be compiled in the build container (no mex.h):
 *   mex -I../include desc_amd_mex.c -L../desc_amd -ldesc_amd
 */
#include <string.h>

#include "mex.h"
#include <math.h>

#include "desc_amd.h"

static void problem_from(const mxArray* ind, const mxArray* rij, desc_problem* p) {
    if (!mxIsInt32(ind) || mxGetN(ind) != 2) mexErrMsgIdAndTxt("desc_amd:Ind", "Ind0 must be m x 2 int32");
    if (!mxIsDouble(rij) || mxIsComplex(rij)) mexErrMsgIdAndTxt("desc_amd:RijMat", "RijMat must be real double");
    const mwSize m = mxGetM(ind);
    if (mxGetNumberOfElements(rij) != 9 * m) mexErrMsgIdAndTxt("desc_amd:RijMat", "RijMat must be 3 x 3 x m");
    const int32_t* d = (const int32_t*)mxGetData(ind);
    int32_t nmax = -1;
    for (mwSize e = 0; e < m; ++e) if (d[m + e] > nmax) nmax = d[m + e];
    p->n = (int64_t)nmax + 1; p->m = (int64_t)m; p->ind_i = d; p->ind_j = d + m; p->rij = mxGetPr(rij);
}

static mxArray* rotations(int64_t n) {
    mwSize dims[3] = {3, 3, (mwSize)n};
    return mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    char cmd[32];
    if (nrhs < 3 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgIdAndTxt("desc_amd:cmd", "first argument: 'spectral' | 'gcw' | 'cemp' | 'refine'");
    desc_problem prob;
    problem_from(prhs[1], prhs[2], &prob);
    int rc = DESC_OK;
    if (!strcmp(cmd, "spectral") || !strcmp(cmd, "gcw")) {
        const int gcw = !strcmp(cmd, "gcw");
        mxArray* w = NULL;
        if (gcw) {
            if (nrhs < 4 || mxGetNumberOfElements(prhs[3]) != (mwSize)prob.m) mexErrMsgIdAndTxt("desc_amd:S", "S_vec must have m entries");
            w = mxCreateDoubleMatrix(1, prob.m, mxREAL);
            const double* s = mxGetPr(prhs[3]);
            for (int64_t e = 0; e < prob.m; ++e) { double x = s[e]; mxGetPr(w)[e] = 1.0 / (x * sqrt(x) + 1e-8); }   /* GCW.m:20 */
        }
        plhs[0] = rotations(prob.n);
        rc = desc_spectral_run(&prob, gcw ? mxGetPr(w) : NULL, gcw, 0.0, 0, 0, mxGetPr(plhs[0]), NULL);
    } else if (!strcmp(cmd, "cemp")) {
        if (nrhs < 6) mexErrMsgIdAndTxt("desc_amd:cemp", "usage: ('cemp', Ind0, RijMat, beta, max_iter, nsample [, seed])");
        plhs[0] = mxCreateDoubleMatrix(1, prob.m, mxREAL);
        rc = desc_cemp_run(&prob, mxGetPr(prhs[3]), (int32_t)mxGetNumberOfElements(prhs[3]), (int32_t)mxGetScalar(prhs[4]),
                           (int32_t)mxGetScalar(prhs[5]), nrhs > 6 ? (uint64_t)mxGetScalar(prhs[6]) : 0, 0, mxGetPr(plhs[0]), NULL);
    } else if (!strcmp(cmd, "refine")) {
        if (nrhs < 5 || mxGetNumberOfElements(prhs[3]) != (mwSize)prob.m || mxGetNumberOfElements(prhs[4]) != (mwSize)(9 * prob.n))
            mexErrMsgIdAndTxt("desc_amd:refine", "usage: ('refine', Ind0, RijMat, S_vec (m), R_init (3x3xn))");
        plhs[0] = rotations(prob.n);
        desc_refine_info info; memset(&info, 0, sizeof info); info.verbose = 1;
        rc = desc_refine_run(&prob, mxGetPr(prhs[3]), mxGetPr(prhs[4]), 0.0, 0, 0, mxGetPr(plhs[0]), &info);
    } else {
        mexErrMsgIdAndTxt("desc_amd:cmd", "unknown command %s", cmd);
    }
    if (rc != DESC_OK) mexErrMsgIdAndTxt("desc_amd:run", "%s", desc_last_error());
}
