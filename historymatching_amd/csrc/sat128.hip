// sat128.hip -- 128x128 fp64 specialisation of the saturation sweep.  (placeholder: not yet applicable)
#include "fwd.h"
int launch_saturation_128(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k) { (void)f; (void)S_in; (void)S_out; (void)S_stride; (void)k; return -1; }
