// update.hip -- ensemble-smoother update (placeholder entry points; filled in next)
#include "common.h"
extern "C" int hm_es_update(hm_ctx*, int, int, int, const void*, const void*, const void*, const void*, const void*, int, void*, hm_stats*) { hm_set_error("hm_es_update: not implemented"); return 3; }
extern "C" int hm_es_update_loc(hm_ctx*, int, int, int, const void*, const void*, const void*, const void*, const void*, const void*, double, int, void*, hm_stats*) { hm_set_error("hm_es_update_loc: not implemented"); return 3; }
extern "C" int hm_upd_create(hm_ctx*, int, int, int, int, int, int, hm_upd**) { hm_set_error("not implemented"); return 3; }
extern "C" void hm_upd_destroy(hm_upd*) {}
extern "C" int hm_upd_set_inputs(hm_upd*, const void*, const void*, const void*, const void*, const void*, const void*, double) { return 3; }
extern "C" int hm_upd_phase(hm_upd*, int) { return 3; }
extern "C" void* hm_upd_reduce_buffer(hm_upd*, int, long long*) { return nullptr; }
extern "C" int hm_upd_sync(hm_upd*, hm_stats*) { return 3; }
extern "C" int hm_upd_get_output(hm_upd*, void*) { return 3; }
extern "C" void* hm_upd_device_ptr(hm_upd*, const char*) { return nullptr; }
