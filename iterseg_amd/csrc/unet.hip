// unet.hip -- placeholder while the watershed stage is brought up on hardware; replaced by
// the real U-Net kernels next.
#include "isx_common.h"
namespace isx { struct UNetState {}; void unet_destroy(UNetState *u) { delete u; } }
extern "C" {
#define NI(...) { isx::set_error("not implemented yet"); return ISX_ERR_STATE; }
int isx_load_weight_host(isx_ctx *, const char *, const void *, const int64_t *, int) NI()
int isx_weights_commit(isx_ctx *, int) NI()
int isx_get_weight_host(isx_ctx *, const char *, void *, int64_t) NI()
int isx_unet_forward_chunks(isx_ctx *, const float *, const int64_t *, const int32_t *, int64_t, const int64_t *, float *) NI()
int isx_predict_volume(isx_ctx *, const float *, const int64_t *, const int32_t *, int64_t, int64_t, const int64_t *, float *) NI()
int isx_unet_tap_host(isx_ctx *, const char *, float *, int64_t, int64_t *) NI()
int isx_unet_set_batch(isx_ctx *, int) NI()
int isx_normalise_volume(isx_ctx *, float *, int64_t) NI()
int isx_segment_volume(isx_ctx *, const float *, const int64_t *, const int64_t *, const int64_t *, float *, uint32_t *) NI()
int isx_segment_volume_host(isx_ctx *, const float *, const int64_t *, const int64_t *, const int64_t *, uint32_t *) NI()
}
