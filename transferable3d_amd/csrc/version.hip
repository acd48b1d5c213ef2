// Identity of the build: ABI version + a hash over the HIP sources and the ABI header the library was compiled from.
//
// transferable3d_amd/build.py passes -DT3D_SOURCE_HASH="<16 hex digits>" (sha256 over csrc/* and include/t3d.h) when it compiles this
// file; abi.load() compares it with the sources lying next to the library and refuses a library that was built from other sources
// (*.so is git-ignored but ships to the GPU box: a stale-but-newer file must not be tested and benchmarked as HEAD).  The marker
// string is also what build.py looks for in the file to decide whether a rebuild is needed.
#include <string.h>
#include "t3d.h"

#ifndef T3D_SOURCE_HASH
#define T3D_SOURCE_HASH "unknown"
#endif

static const char t3d_source_hash_marker[] = "T3D_SOURCE_HASH=" T3D_SOURCE_HASH;

extern "C" int t3d_source_hash(char* out, int cap) {
  const char* h = t3d_source_hash_marker + 16;
  const int n = (int)strlen(h);
  if (!out || cap <= n) return T3D_ERR_ARG;
  memcpy(out, h, (size_t)n + 1);
  return T3D_OK;
}
