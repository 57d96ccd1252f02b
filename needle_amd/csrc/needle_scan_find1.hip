// Instantiations of the tiled scan kernel for find() on 8-bit rows (DFAClassBuilder.createFindMethodInternal :625-659,
// createIndexMethod :335-471, createIndexMethodReversed :529-586).
#include "needle_scan.h"
namespace needle {
hipError_t launch_scan_find1(const ScanArgs &a, bool guard, LaunchShape sh, hipStream_t s) { return launch_m<OP_FIND, 1>(a, guard, sh, s); }
} // namespace needle
