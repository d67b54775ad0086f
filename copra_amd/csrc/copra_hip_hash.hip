// copra_hip_hash.hip -- the hash of the library's sources, compiled in (Makefile: COPRA_SRC_HASH; the same formula as copra_amd/_capi.py::
// source_hash).  A translation unit of its own, rebuilt whenever any source changes: it keys the cache of run-time-compiled kernels
// (copra_hip_jit.hip) and ties committed rocprofv3 summaries to the build they were taken on (bench.py: roofline.traffic_stale).
#include "../../include/copra_hip.h"
#ifndef COPRA_SRC_HASH
#error "build through copra_amd/csrc/Makefile (it defines COPRA_SRC_HASH)"
#endif
extern "C" const char* copra_source_hash(void) { return COPRA_SRC_HASH; }
