// encoder.hip — placeholder entry points (replaced by the real encoder below in this round)
#include "otter_common.h"
using namespace otter;
struct otter_encoder { int dummy; };
extern "C" {
int otter_encoder_create(const otter_encoder_cfg*, int, otter_encoder** out) { if (out) *out = nullptr; return set_err(OTTER_ENOTIMPL, "encoder not built yet"); }
int otter_encoder_set_weight(otter_encoder*, const char*, const float*, const int64_t*, int) { return set_err(OTTER_ENOTIMPL, "encoder not built yet"); }
int otter_encoder_finalize(otter_encoder*, int) { return set_err(OTTER_ENOTIMPL, "encoder not built yet"); }
int otter_encoder_forward(otter_encoder*, const int64_t*, const int64_t*, const int64_t*, const int64_t*, const int64_t*, int, int, float*, void*) { return set_err(OTTER_ENOTIMPL, "encoder not built yet"); }
int otter_encoder_forward_hidden(otter_encoder*, const int64_t*, const int64_t*, const int64_t*, int, int, int, float*, void*) { return set_err(OTTER_ENOTIMPL, "encoder not built yet"); }
int otter_encoder_out_dim(const otter_encoder*) { return -1; }
void otter_encoder_destroy(otter_encoder*) {}
}
