// kernel instantiations of libflowsim_hip.so, part "nodiag" (see fs_entries.hpp)
#include "fs_entries.hpp"

FS_LIST_NODIAG(FS_INSTANTIATE_NODIAG)
FS_LIST_TAIL(FS_INSTANTIATE_TAIL)
