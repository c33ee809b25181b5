/* TEST MOCK -- not MATLAB's header.  Prototypes of the handful of C Matrix / MEX API functions that
 * matlab/desc_pgd_mex.c and matlab/desc_amd_mex.c call, as MathWorks documents them, so that the
 * shims can be SYNTAX-checked (gcc -fsyntax-only) in an image without MATLAB.  Nothing links
 * against this; the real build uses MATLAB's own mex.h. */
#ifndef DESC_TEST_MOCK_MEX_H
#define DESC_TEST_MOCK_MEX_H
#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxUNKNOWN_CLASS = 0, mxDOUBLE_CLASS = 6, mxINT32_CLASS = 12 } mxClassID;
double* mxGetPr(const mxArray*);
void* mxGetData(const mxArray*);
double mxGetScalar(const mxArray*);
size_t mxGetM(const mxArray*);
size_t mxGetN(const mxArray*);
size_t mxGetNumberOfElements(const mxArray*);
mwSize mxGetNumberOfDimensions(const mxArray*);
const mwSize* mxGetDimensions(const mxArray*);
bool mxIsDouble(const mxArray*);
bool mxIsInt32(const mxArray*);
bool mxIsComplex(const mxArray*);
bool mxIsEmpty(const mxArray*);
bool mxIsStruct(const mxArray*);
mxArray* mxGetField(const mxArray*, mwIndex, const char*);
void mxSetField(mxArray*, mwIndex, const char*, mxArray*);
int mxGetString(const mxArray*, char*, mwSize);
mxArray* mxCreateDoubleMatrix(mwSize, mwSize, mxComplexity);
mxArray* mxCreateDoubleScalar(double);
mxArray* mxCreateNumericArray(mwSize, const mwSize*, mxClassID, mxComplexity);
mxArray* mxCreateStructMatrix(mwSize, mwSize, int, const char**);
void mxSetN(mxArray*, mwSize);
int mxSetDimensions(mxArray*, const mwSize*, mwSize);
void mexErrMsgIdAndTxt(const char*, const char*, ...);
int mexPrintf(const char*, ...);
int mexAtExit(void (*exit_fcn)(void));
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
#endif
