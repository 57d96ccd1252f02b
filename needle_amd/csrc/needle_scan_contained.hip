// Instantiations of the tiled scan kernel for containedIn() (DFAClassBuilder.createContainedInMethod :956-1025).
#include "needle_scan.h"
namespace needle {
hipError_t launch_scan_contained_in(const ScanArgs &a, int cw, bool guard, LaunchShape sh, hipStream_t s) {
    return launch_c<OP_CONTAINED_IN>(a, cw, guard, sh, s);
}
} // namespace needle
