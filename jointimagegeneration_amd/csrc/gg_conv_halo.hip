// Halo-tile fast path for 3x3(x3) stride-1 convolutions (placeholder until the tuned kernel lands).
#include "gg_common.h"
struct ConvParams;
int gg_conv_halo_try(const ConvParams &, hipStream_t) { return GG_ERR_UNSUPPORTED; }
