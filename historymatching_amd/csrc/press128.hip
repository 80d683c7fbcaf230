// press128.hip -- 128-wide (Ny = 128) fp64 specialisation of the pressure step.  (placeholder: not yet applicable)
#include "fwd.h"
int launch_pressure_128(hm_fwd* f, const void* S, long long S_stride, int k) { (void)f; (void)S; (void)S_stride; (void)k; return -1; }
