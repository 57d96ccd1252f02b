// Regex -> reference-layout tables (the role DFACompiler.compileToBytes plays up to the point where it
// hands four DFAs to DFAClassBuilder: needle-compiler/.../DFACompiler.java:45-65).
#pragma once
#include <string>
#include "needle_lower.h"

namespace needle {

// Returns a NEEDLE_* status (include/needle_hip.h); on failure `err` holds the message.
int compile_regex(const std::u16string &regex, int flags, RefTables &out, std::string &err);

} // namespace needle
