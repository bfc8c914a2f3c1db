/* Prototype-only stand-in for MATLAB's mex.h, used by tools/check_mex_syntax.sh to run
 * `g++ -fsyntax-only` over the gateways under ofdm-course_amd/mex in an environment without MATLAB.
 * It declares the handful of MEX API names the gateways use; nothing here is ever linked or run,
 * and it is NOT MATLAB's header (signatures follow the public R2018a C Matrix API docs). */
#pragma once
#include <stddef.h>
typedef struct mxArray_tag mxArray;
typedef bool mxLogical;
typedef struct { double real, imag; } mxComplexDouble;
typedef enum { mxREAL, mxCOMPLEX } mxComplexity;
extern "C" {
size_t mxGetM(const mxArray*);
size_t mxGetN(const mxArray*);
size_t mxGetNumberOfElements(const mxArray*);
bool mxIsDouble(const mxArray*);
bool mxIsComplex(const mxArray*);
bool mxIsLogical(const mxArray*);
bool mxIsChar(const mxArray*);
bool mxIsClass(const mxArray*, const char*);
double* mxGetDoubles(const mxArray*);
mxComplexDouble* mxGetComplexDoubles(const mxArray*);
mxLogical* mxGetLogicals(const mxArray*);
double mxGetScalar(const mxArray*);
char* mxArrayToString(const mxArray*);
void mxFree(void*);
void mxDestroyArray(mxArray*);
mxArray* mxCreateDoubleMatrix(size_t, size_t, mxComplexity);
mxArray* mxCreateDoubleScalar(double);
int mexCallMATLAB(int, mxArray*[], int, mxArray*[], const char*);
void mexErrMsgIdAndTxt(const char*, const char*, ...);
void mexWarnMsgIdAndTxt(const char*, const char*, ...);
int mexAtExit(void (*)(void));
void mexFunction(int, mxArray*[], int, const mxArray*[]);
}
