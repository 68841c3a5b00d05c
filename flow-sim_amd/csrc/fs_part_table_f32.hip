// kernel instantiations of libflowsim_hip.so, part "table_f32" (see fs_entries.hpp)
#include "fs_entries.hpp"

FS_LIST_TABLE(FS_INSTANTIATE, float, FS_F32)
