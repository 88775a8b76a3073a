// temporary: replaced by engine.cpp
#include "common.h"
extern "C" {
int dzn_create(const dzn_config*, dzn_handle**) { return DZN_E_STATE; }
int dzn_load_tensor(dzn_handle*, const char*, const void*, const int64_t*, int32_t, int32_t) { return DZN_E_STATE; }
int dzn_finalize_weights(dzn_handle*) { return DZN_E_STATE; }
int dzn_num_frames(const dzn_handle*, int32_t) { return DZN_E_STATE; }
int dzn_segment_forward(dzn_handle*, const float*, int32_t, int32_t, float*, uint8_t*, void*) { return DZN_E_STATE; }
int dzn_embed_forward(dzn_handle*, const float*, const float*, int32_t, int32_t, int32_t, int32_t, float*, void*) { return DZN_E_STATE; }
int dzn_debug_fetch(dzn_handle*, const char*, float*, int64_t, int64_t*) { return DZN_E_STATE; }
int dzn_num_ignored(const dzn_handle*) { return 0; }
int64_t dzn_workspace_bytes(const dzn_handle*) { return 0; }
const char* dzn_last_error(const dzn_handle*) { return ""; }
int dzn_destroy(dzn_handle*) { return 0; }
const char* dzn_version(void) { return "dzn-hip 0.1 (gfx950)"; }
}
