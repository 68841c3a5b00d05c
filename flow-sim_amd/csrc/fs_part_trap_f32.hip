// kernel instantiations of libflowsim_hip.so, part "trap_f32" (see fs_entries.hpp)
#include "fs_entries.hpp"

FS_LIST_TRAP(FS_INSTANTIATE, float, FS_F32)
