// Host-side plumbing shared by all entry points: ABI version and the per-thread
// last-error string.
#include "common.h"
#include <string.h>

static thread_local char g_err[256] = "";

void hi3d_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

extern "C" int hi3d_abi_version(void) { return HI3D_ABI_VERSION; }
extern "C" const char* hi3d_last_error(void) { return g_err; }
