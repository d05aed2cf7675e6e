/* tests/r_api_stub/R_ext/Random.h -- see ../R.h: declarations only */
#ifndef BWGR_R_API_STUB_RANDOM_H
#define BWGR_R_API_STUB_RANDOM_H
void GetRNGstate(void);
void PutRNGstate(void);
double unif_rand(void);
#endif
