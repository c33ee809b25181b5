/* desc_amd_mex.c -- MEX dispatcher for the "next" rows of the C ABI (include/desc_amd.h):
 *
 *   R     = desc_amd_mex('spectral', Ind0, RijMat)                       desc_spectral_run, Spectral.m
 *   R     = desc_amd_mex('gcw',      Ind0, RijMat, S_vec)                desc_spectral_run, GCW.m
 *   SVec  = desc_amd_mex('cemp',     Ind0, RijMat, beta, max_iter, nsample, seed)   desc_cemp_run, CEMP.m
 *   R     = desc_amd_mex('refine',   Ind0, RijMat, S_vec, R_init)        desc_refine_run, DESC.m:265-313
 *   [R_est, R_init, S_vec, info] = desc_amd_mex('desc', Ind0, RijMat, opt)   the whole DESC.m:16-313 on ONE device-resident
 *                                                      problem (desc_problem_upload): PGD -> GCW -> refinement
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

static void mex_progress(void* user, int32_t it, double avg, double obj) {     /* DESC_PGD.m:241 */
    (void)user;
    mexPrintf("iter %d: average change in S_vec %f, objective value: %f\n", (int)it, avg, obj);
}
static double field_or(const mxArray* s, const char* name, double dflt) {
    const mxArray* f = mxGetField(s, 0, name);
    return (f && !mxIsEmpty(f)) ? mxGetScalar(f) : dflt;
}

static mxArray* rotations(int64_t n) {
    mwSize dims[3] = {3, 3, (mwSize)n};
    return mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
}

/* the library parks device and host blocks between calls (DESC_CACHE_MB / DESC_HOST_CACHE_MB): give them back when MATLAB
 * clears the MEX file or exits */
static void release_parked_blocks(void) { (void)desc_trim_memory(); }
static int at_exit_registered = 0;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (!at_exit_registered) { mexAtExit(release_parked_blocks); at_exit_registered = 1; }
    char cmd[32];
    if (nrhs < 3 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgIdAndTxt("desc_amd:cmd", "first argument: 'spectral' | 'gcw' | 'cemp' | 'refine' | 'desc'");
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
    } else if (!strcmp(cmd, "desc")) {
        /* constant / piecewise / decayed-plain steps; the Adam plugin (per-cycle state across calls) goes through
         * desc_pgd_mex + 'gcw' + 'refine' */
        if (nrhs < 4 || !mxIsStruct(prhs[3])) mexErrMsgIdAndTxt("desc_amd:desc", "usage: ('desc', Ind0, RijMat, opt)");
        desc_params p;
        desc_params_default(&p);
        p.iters = (int32_t)field_or(prhs[3], "iters", 100);
        p.step_kind = (int32_t)field_or(prhs[3], "step_kind", 0);
        p.lr = field_or(prhs[3], "lr", 0.01);
        p.decay_interval = field_or(prhs[3], "decay_interval", 25);
        p.hybrid_strategy = (int32_t)field_or(prhs[3], "hybrid_strategy", 0);
        p.t0 = (int32_t)field_or(prhs[3], "t0", 0);
        p.seed = (uint64_t)field_or(prhs[3], "seed", 0);
        p.device = (int32_t)field_or(prhs[3], "device", 0);
        if (p.step_kind == DESC_STEP_HYBRID && p.hybrid_strategy == 0) mexErrMsgIdAndTxt("desc_amd:desc", "Adam state is carried by desc_pgd_mex");
        if (field_or(prhs[3], "verbose", 1) != 0) p.progress = mex_progress;
        desc_device_problem* dp = NULL; desc_structure* st = NULL; desc_pgd* h = NULL;
        plhs[0] = rotations(prob.n);
        mxArray* R_init = rotations(prob.n);
        mxArray* S = mxCreateDoubleMatrix(1, prob.m, mxREAL);
        mxArray* obj = mxCreateDoubleMatrix(1, p.iters > 0 ? p.iters : 1, mxREAL);
        mxArray* avg = mxCreateDoubleMatrix(1, p.iters > 0 ? p.iters : 1, mxREAL);
        desc_result r; memset(&r, 0, sizeof r);
        r.s_vec = mxGetPr(S); r.obj_trace = mxGetPr(obj); r.avg_change_trace = mxGetPr(avg);
        desc_refine_info rinfo; memset(&rinfo, 0, sizeof rinfo); rinfo.verbose = 1;
        rc = desc_problem_upload(&prob, p.device, &dp);
        if (rc == DESC_OK) {
            rc = desc_structure_build(&prob, p.n_sample_min, p.seed, DESC_BUILD_DEVICE, p.device, &st);
            if (rc == DESC_ERR_TOO_LARGE) rc = desc_structure_build(&prob, p.n_sample_min, p.seed, DESC_BUILD_HOST, p.device, &st);
        }
        if (rc == DESC_OK) rc = desc_pgd_create_dev(dp, st, 0, 1, &h);
        if (st) desc_structure_free(st);
        if (rc == DESC_OK) rc = desc_pgd_run(h, &p, &r);                                     /* DESC.m:16-261 */
        if (h) desc_pgd_destroy(h);
        if (rc == DESC_OK) rc = desc_gcw_run_dev(dp, mxGetPr(S), 0.0, 0, mxGetPr(R_init), NULL);   /* DESC.m:263 */
        if (rc == DESC_OK) {
            mexPrintf("Rotation Initialized!\nStart DESC refinement ...\n");                /* DESC.m:283-284 */
            rc = desc_refine_run_dev(dp, mxGetPr(S), mxGetPr(R_init), 0.0, 0, mxGetPr(plhs[0]), &rinfo);   /* DESC.m:265-313 */
        }
        if (dp) desc_problem_free(dp);
        if (nlhs > 1) plhs[1] = R_init;
        if (nlhs > 2) plhs[2] = S;
        if (nlhs > 3) {
            const char* names[] = {"iters_run", "t_end", "obj_vals", "avg_change", "refine_iters"};
            plhs[3] = mxCreateStructMatrix(1, 1, 5, names);
            mxSetField(plhs[3], 0, "iters_run", mxCreateDoubleScalar(r.iters_run));
            mxSetField(plhs[3], 0, "t_end", mxCreateDoubleScalar(r.t_end));
            mxSetField(plhs[3], 0, "obj_vals", obj);
            mxSetField(plhs[3], 0, "avg_change", avg);
            mxSetField(plhs[3], 0, "refine_iters", mxCreateDoubleScalar(rinfo.iters));
        }
    } else {
        mexErrMsgIdAndTxt("desc_amd:cmd", "unknown command %s", cmd);
    }
    if (rc != DESC_OK) mexErrMsgIdAndTxt("desc_amd:run", "%s", desc_last_error());
}
