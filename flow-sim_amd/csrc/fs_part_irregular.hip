// kernel instantiations of libflowsim_hip.so, part "irregular" (see fs_entries.hpp)
#include "fs_entries.hpp"

FS_LIST_IRREGULAR(FS_INSTANTIATE)
