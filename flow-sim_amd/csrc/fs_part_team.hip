// kernel instantiations of libflowsim_hip.so, part "team" (see fs_entries.hpp)
#include "fs_entries.hpp"

FS_LIST_TEAM(FS_INSTANTIATE_TEAM)
