#pragma once
#include <hip/hip_runtime.h>
// kernel classes for urn_prof_read()
#define URN_PROF_GCONV 0   /* k_gconv_fwd<MB,NB>: forward and input-gradient gather conv */
#define URN_PROF_DW 1      /* k_gconv_dw: weight gradient */
#define URN_PROF_INTEGER 2 /* integer phase: urn_sites_build, urn_level_down_tables, urn_rulebook_subm_multi (one record per call) */
bool urn_prof_on();
void urn_prof_begin(int kind, hipStream_t st);
void urn_prof_end(hipStream_t st);
