// kernel instantiations of libflowsim_hip.so, part "ensemble" (see fs_entries.hpp)
#include "fs_entries.hpp"

FS_LIST_ENSEMBLE(FS_INSTANTIATE_ENS)
