// ofx_fused.hip -- FUSED engine (placeholder until the LDS-resident FFT kernel lands)
#include "ofx_common.h"
bool ofx_fused_supported(int) { return false; }
int ofx_fused_prepare_slot(ofx_plan*, int, const double*) { return OFX_OK; }
int ofx_fused_process(ofx_plan*, const float*, const uint8_t*, long long, float*, hipStream_t) {
    ofx_set_error("FUSED engine not built"); return OFX_ERR_UNSUPPORTED; }
int ofx_fused_release(ofx_plan*) { return OFX_OK; }
