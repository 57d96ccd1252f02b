#include "needle_regex.h"
#include "../../include/needle_hip.h"

namespace needle {

int compile_regex(const std::u16string &, int, RefTables &, std::string &err) {
    err = "regex compiler not built yet";
    return NEEDLE_ERR_UNSUPPORTED;
}

} // namespace needle
