// kernel instantiations of libflowsim_hip.so, part "rect_f32" (see fs_entries.hpp)
#include "fs_entries.hpp"

FS_LIST_RECT(FS_INSTANTIATE, float, FS_F32)
