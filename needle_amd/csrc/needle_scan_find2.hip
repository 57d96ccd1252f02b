// Instantiations of the tiled scan kernel for find() on UTF-16 rows; the dispatch of find() by char width.
#include "needle_scan.h"
namespace needle {
hipError_t launch_scan_find1(const ScanArgs &a, bool guard, LaunchShape sh, hipStream_t s);
hipError_t launch_scan_find(const ScanArgs &a, int cw, bool guard, LaunchShape sh, hipStream_t s) {
    return cw == 1 ? launch_scan_find1(a, guard, sh, s) : launch_m<OP_FIND, 2>(a, guard, sh, s);
}
} // namespace needle
