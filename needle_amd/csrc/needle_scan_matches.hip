// Instantiations of the tiled scan kernel for matches() (DFAClassBuilder.createMatchesMethod :854-912).
#include "needle_scan.h"
namespace needle {
hipError_t launch_scan_matches(const ScanArgs &a, int cw, bool guard, LaunchShape sh, hipStream_t s) {
    return launch_c<OP_MATCHES>(a, cw, guard, sh, s);
}
} // namespace needle
