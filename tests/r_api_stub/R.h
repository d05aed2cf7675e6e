/* tests/r_api_stub/R.h -- TEST INFRASTRUCTURE ONLY: declarations (no definitions) of the few names of R's C API that rshim/bwgr_shim.c uses, as
 * documented in "Writing R Extensions", so that the shim -- this repository's own code -- can be SYNTAX-checked (gcc -fsyntax-only) in an image
 * without R.  It pins nothing and builds nothing: the shim is compiled for real only where R's own headers exist. */
#ifndef BWGR_R_API_STUB_R_H
#define BWGR_R_API_STUB_R_H
#include <stddef.h>
char *R_alloc(size_t n, int size);
void Rf_error(const char *fmt, ...) __attribute__((noreturn));
#endif
