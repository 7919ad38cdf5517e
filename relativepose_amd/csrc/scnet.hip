// placeholder until the conv stack lands (next commit): entry points exist so the library links.
#include "common.h"
extern "C" {
RelposeSCNet* relpose_scnet_create(int32_t, int32_t) { return nullptr; }
void relpose_scnet_destroy(RelposeSCNet*) {}
int relpose_scnet_set_param(RelposeSCNet*, const char*, const float*, size_t) { return RELPOSE_EINVAL; }
int relpose_scnet_finalize(RelposeSCNet*) { return RELPOSE_EINVAL; }
int64_t relpose_scnet_num_params(const RelposeSCNet*) { return 0; }
size_t relpose_scnet_workspace_bytes(const RelposeSCNet*, int32_t, int32_t, int32_t) { return 0; }
int relpose_scnet_forward(RelposeSCNet*, const float*, float*, int32_t, int32_t, int32_t, void*, size_t, void*) { return RELPOSE_EINVAL; }
int64_t relpose_scnet_read_tap(RelposeSCNet*, const char*, float*, void*, void*) { return -1; }
int relpose_scnet_profile(RelposeSCNet*, const float*, float*, int32_t, int32_t, int32_t, void*, size_t, int32_t, double*, double*, int64_t*, void*) { return RELPOSE_EINVAL; }
}
