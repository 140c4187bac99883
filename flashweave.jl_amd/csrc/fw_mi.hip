// FlashWeave-F / HE-F (discrete, contingency-table) device path -- TEMPORARY STUB, replaced below in this round.
#include "fw_internal.h"
int fwi_mi_upload(fw_ctx *ctx, const int64_t *, const int32_t *, const int32_t *) { return fw_fail(ctx, FW_ERR_STATE, "discrete path not built yet"); }
int fwi_mi_level0(fw_ctx *ctx, std::vector<int32_t> &, std::vector<int32_t> &, std::vector<double> &, std::vector<double> &, int64_t *) { return fw_fail(ctx, FW_ERR_STATE, "discrete path not built yet"); }
int fwi_mi_test_batch(fw_ctx *ctx, int64_t, const int32_t *, const int32_t *, const int64_t *, const int32_t *, fw_test_result *) { return fw_fail(ctx, FW_ERR_STATE, "discrete path not built yet"); }
int fwi_mi_segments(fw_ctx *ctx, int64_t, const FwSeg *, const int32_t *, FwSegOut *) { return fw_fail(ctx, FW_ERR_STATE, "discrete path not built yet"); }
