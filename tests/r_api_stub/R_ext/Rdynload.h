/* tests/r_api_stub/R_ext/Rdynload.h -- see ../R.h: declarations only */
#ifndef BWGR_R_API_STUB_RDYNLOAD_H
#define BWGR_R_API_STUB_RDYNLOAD_H
#include "../Rinternals.h"
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
typedef struct R_CMethodDef_ R_CMethodDef;
typedef struct R_FortranMethodDef_ R_FortranMethodDef;
typedef struct R_ExternalMethodDef_ R_ExternalMethodDef;
int R_registerRoutines(DllInfo *info, const R_CMethodDef *const croutines, const R_CallMethodDef *const callRoutines,
                       const R_FortranMethodDef *const fortranRoutines, const R_ExternalMethodDef *const externalRoutines);
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value);
#endif
