// placeholder until the conv engine lands
#include "../../include/dif.h"
#include "dif_internal.hpp"
using namespace dif;
extern "C" {
int dif_net_create(dif_net**, const char*, const char*, int, int, int) { return set_error("not built yet"); }
int dif_net_destroy(dif_net*) { return 0; }
int dif_net_param_count(const dif_net*) { return 0; }
int dif_net_param_info(const dif_net*, int, const char**, int*, int64_t*) { return set_error("not built yet"); }
int dif_net_set_param(dif_net*, const char*, const float*, int64_t) { return set_error("not built yet"); }
int dif_net_get_param(const dif_net*, const char*, float*, int64_t) { return set_error("not built yet"); }
int dif_net_set_input_transform(dif_net*, float, const float*, int) { return set_error("not built yet"); }
int dif_net_finalize(dif_net*, int) { return set_error("not built yet"); }
int dif_net_output_dim(const dif_net*, int64_t*) { return set_error("not built yet"); }
int dif_net_embed(dif_net*, const void*, int, int, int, float*, void*) { return set_error("not built yet"); }
double dif_net_flops_per_image(const dif_net*) { return 0; }
int dif_net_launch_count(const dif_net*) { return 0; }
int dif_arcmargin_create(dif_arcmargin**, int, int64_t, float, float) { return set_error("not built yet"); }
int dif_arcmargin_destroy(dif_arcmargin*) { return 0; }
int dif_arcmargin_set_weight(dif_arcmargin*, const float*, void*) { return set_error("not built yet"); }
int dif_arcmargin_logits(dif_arcmargin*, const float*, const int64_t*, int, float*, void*) { return set_error("not built yet"); }
}
